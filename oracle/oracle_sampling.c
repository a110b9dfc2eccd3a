/*
 * oracle_sampling.c — CPU restatement of the seeded samplers of the GPU path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.c).  PARITY UNPINNED at the byte level for
 * the same reason as oracle.c; the ChaCha20 block function is pinned by the RFC 8439
 * §2.3.2 known-answer vector (tests/test_oracle_sampling.py), everything else by the
 * reference's own acceptance predicates (G*z = v, A*x = u, norm bounds, window
 * invariance) and by distribution moments.
 *
 * The reference's CPU path draws from OpenFHE's generators with OS entropy and is not
 * reproducible; what IS specified in-tree is the device sampler, which this file
 * restates (paths relative to /root/reference):
 *   - stream keying: HChaCha20 sub-key from (seed, domain tag, stream2) as in
 *     cuda/src/ChaCha.cu:104-167; block counter starts at 0 and (stream0, stream1) form the
 *     96-bit nonce (the reference's counter = stream0 lets adjacent streams share keystream;
 *     mxx_amd/csrc/rng.h states the layout)
 *   - uniform-mod rejection, bit, ternary, Karney's exact discrete Gaussian with the
 *     reference's iteration caps                  cuda/src/matrix/MatrixSampling.cu:6-330
 *   - G-lattice sampler (Genise-Micciancio, arbitrary base)
 *                                          cuda/src/matrix/MatrixTrapdoor.cu:701-833
 *   - p1 perturbation: per-coefficient covariance factorisation + conditional sampling
 *                                          cuda/src/matrix/MatrixTrapdoor.cu:95-277
 * Build with -ffp-contract=off: Karney's sampler uses only IEEE add/mul/div/ceil and
 * compares, so integer outputs are bit-identical to the device kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Box-Muller's log / cos as fixed IEEE operation sequences: the same text the device compiles */
#include "../mxx_amd/csrc/detmath.h"

double orc_det_log(double x) { return det_log(x); }
double orc_det_cos2pi(double u) { return det_cos2pi(u); }

typedef struct {
    uint32_t state[16];
    uint32_t block[16];
    uint32_t pos; /* next 16-bit draw of the current block, 0..32 */
} rng_t;

static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d)                 \
    a += b; d ^= a; d = rotl(d, 16);   \
    c += d; b ^= c; b = rotl(b, 12);   \
    a += b; d ^= a; d = rotl(d, 8);    \
    c += d; b ^= c; b = rotl(b, 7);

static void rounds20(uint32_t x[16]) {
    for (int i = 0; i < 10; ++i) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
}

/* RFC 8439 block function: out = rounds(in) + in */
void orc_chacha20_block(const uint32_t in[16], uint32_t out[16]) {
    uint32_t x[16];
    memcpy(x, in, sizeof(x));
    rounds20(x);
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

static void rng_init(rng_t *r, const uint64_t seed[4], uint64_t s0, uint64_t s1, uint64_t s2, uint64_t tag) {
    uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    for (int i = 0; i < 4; ++i) {
        x[4 + 2 * i] = (uint32_t)seed[i];
        x[5 + 2 * i] = (uint32_t)(seed[i] >> 32);
    }
    x[12] = (uint32_t)tag; x[13] = (uint32_t)(tag >> 32);
    x[14] = (uint32_t)s2;  x[15] = (uint32_t)(s2 >> 32);
    rounds20(x); /* HChaCha20 */
    r->state[0] = 0x61707865u; r->state[1] = 0x3320646eu; r->state[2] = 0x79622d32u; r->state[3] = 0x6b206574u;
    r->state[4] = x[0]; r->state[5] = x[1]; r->state[6] = x[2]; r->state[7] = x[3];
    r->state[8] = x[12]; r->state[9] = x[13]; r->state[10] = x[14]; r->state[11] = x[15];
    /* pure 32-bit block counter; the stream words live in the 96-bit nonce (rng.h explains the
       departure from cuda/src/ChaCha.cu:138-149, whose counter = stream0 makes adjacent streams overlap) */
    r->state[12] = 0; r->state[13] = (uint32_t)s0;
    r->state[14] = (uint32_t)s1;
    r->state[15] = ((uint32_t)(s0 >> 32) & 0xffffu) | ((uint32_t)(s1 >> 32) << 16);
    r->pos = 32;
}

/* The keystream is consumed as little-endian 16-bit draws, 32 per ChaCha20 block (mxx_amd/csrc/rng.h: Karney's
 * uniform deviates are only ever compared, so 16 bits decide all but one comparison in 65536; a 64-bit word is four
 * consecutive draws, i.e. the same bytes as before whenever it starts on a multiple of four draws). */
static uint32_t rng_draw16(rng_t *r) {
    if (r->pos >= 32) {
        orc_chacha20_block(r->state, r->block);
        ++r->state[12];
        r->pos = 0;
    }
    const uint32_t w = r->block[r->pos >> 1];
    const uint32_t v = (r->pos & 1u) ? (w >> 16) : (w & 0xffffu);
    ++r->pos;
    return v;
}

static uint64_t rng_u64(rng_t *r) {
    uint64_t v = 0;
    for (int i = 0; i < 4; ++i) v |= (uint64_t)rng_draw16(r) << (16 * i);
    return v;
}

/* first `count` u64 words of the stream (seed, s0, s1, s2, tag) */
void orc_rng_stream(const uint64_t *seed, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t tag, uint64_t *out, size_t count) {
    rng_t r;
    rng_init(&r, seed, s0, s1, s2, tag);
    for (size_t i = 0; i < count; ++i) out[i] = rng_u64(&r);
}

static double u01(rng_t *r) {
    const double scale = 1.0 / 9007199254740992.0;
    double u = (double)(rng_u64(r) >> 11) * scale;
    if (u <= 0.0) u = scale;
    else if (u >= 1.0) u = 1.0 - scale;
    return u;
}

static double std_normal(rng_t *r) {
    double u1 = u01(r), u2 = u01(r);
    return sqrt(-2.0 * det_log(u1)) * det_cos2pi(u2);
}


/* Karney, "Sampling exactly from the normal distribution" (arXiv:1303.6257), algorithm D, with the reference's
 * iteration caps (cuda/src/matrix/MatrixSampling.cu:30-147).  A uniform deviate is m 2^-53 with the integer
 * m = max(hi 2^37 + lo, 1): hi is one 16-bit draw, lo (37 bits: three more draws, d1 2^21 + d2 2^5 + (d3 >> 11)) is drawn
 * only when a comparison ties on hi - the threshold's lo first if it is itself a deviate whose lo is still undrawn, then
 * the deviate's.  Every comparison is therefore the one the reference makes on 53-bit deviates ("u < t" on doubles ==
 * "m < ceil(t 2^53)" on integers); only the keystream spent on it differs. */
typedef struct {
    uint32_t hi;  /* up to 65536 for the constant 1.0 */
    uint64_t lo;
    int known;
} lazy_t;

static unsigned long long g_karney_ties; /* test hook: comparisons that needed the low bits */
unsigned long long orc_karney_ties(void) { return g_karney_ties; }

static lazy_t lz_const(double t) { /* t in [0, 1] */
    const uint64_t ticks = (uint64_t)ceil(t * 9007199254740992.0);
    lazy_t v = {(uint32_t)(ticks >> 37), ticks & ((1ull << 37) - 1), 1};
    return v;
}
static lazy_t lz_draw(rng_t *r) {
    lazy_t v = {rng_draw16(r), 0, 0};
    return v;
}
static void lz_materialise(rng_t *r, lazy_t *v) {
    if (v->known) return;
    const uint64_t d1 = rng_draw16(r), d2 = rng_draw16(r), d3 = rng_draw16(r);
    uint64_t lo = (d1 << 21) | (d2 << 5) | (d3 >> 11);
    if (v->hi == 0 && lo == 0) lo = 1; /* the deviate is never 0 */
    v->lo = lo;
    v->known = 1;
}
static int lz_less(rng_t *r, lazy_t *u, lazy_t *t) {
    if (u->hi < t->hi) return 1;
    if (u->hi > t->hi) return 0;
#pragma omp atomic
    ++g_karney_ties;
    lz_materialise(r, t);
    lz_materialise(r, u);
    return u->lo < t->lo;
}

static int k_h(rng_t *r) {
    lazy_t half = lz_const(0.5);
    lazy_t a = lz_draw(r);
    if (!lz_less(r, &a, &half)) return 1;
    for (;;) {
        lazy_t b = lz_draw(r);
        if (!lz_less(r, &b, &a)) return 0;
        a = lz_draw(r);
        if (!lz_less(r, &a, &b)) return 1;
    }
}
static int32_t k_g(rng_t *r) {
    int32_t n = 0;
    while (k_h(r)) { ++n; if (n > 1024) break; }
    return n;
}
static int k_p(rng_t *r, int32_t n) {
    while (n-- && k_h(r)) {}
    return n < 0;
}
static int k_b(rng_t *r, int32_t k, double x) {
    lazy_t y = lz_const(x);
    lazy_t thr = lz_const((2.0 * (double)k + x) / (double)(2 * k + 2));
    int32_t n = 0;
    for (;; ++n) {
        lazy_t z = lz_draw(r);
        if (!lz_less(r, &z, &y)) break;
        lazy_t rr = lz_draw(r);
        if (!lz_less(r, &rr, &thr)) break;
        y = z;
        if (n > 4096) break;
    }
    return (n % 2) == 0;
}
static int64_t karney(rng_t *r, double mean, double stddev) {
    if (!(stddev > 0.0) || !isfinite(mean) || !isfinite(stddev)) return (int64_t)llround(mean);
    const int64_t ceil_std = (int64_t)ceil(stddev);
    if (ceil_std <= 0) return (int64_t)llround(mean);
    for (int iter = 0; iter < (1 << 16); ++iter) {
        int32_t k = k_g(r);
        if (!k_p(r, k * (k - 1))) continue;
        int64_t s = (rng_draw16(r) & 1u) ? 1 : -1;
        double di0 = stddev * (double)k + (double)s * mean;
        int64_t i0 = (int64_t)ceil(di0);
        double x0 = ((double)i0 - di0) / stddev;
        int64_t j = (int64_t)(rng_u64(r) % (uint64_t)ceil_std);
        double x = x0 + (double)j / stddev;
        if (!(x < 1.0) || (x == 0.0 && s < 0 && k == 0)) continue;
        int32_t h = k + 1;
        while (h-- > 0 && k_b(r, k, x)) {}
        if (h >= 0) continue;
        return s * (i0 + j);
    }
    return (int64_t)llround(mean + stddev * std_normal(r));
}

static uint64_t signed_mod(int64_t v, uint64_t q) {
    if (v >= 0) return (uint64_t)v % q;
    uint64_t mag = (uint64_t)(-(v + 1)) + 1;
    uint64_t rem = mag % q;
    return rem == 0 ? 0 : q - rem;
}

static int64_t centered(uint64_t v, uint64_t q) {
    uint64_t half = q >> 1;
    return v <= half ? (int64_t)v : -(int64_t)(q - v);
}

/* `count` Karney samples from one stream (stream0 = s0, stream1 = 1, tag = Gauss) */
void orc_karney(const uint64_t *seed, uint64_t s0, double mean, double stddev, int64_t *out, size_t count) {
    rng_t r;
    rng_init(&r, seed, s0, 1, 0, 0x6f70656e66686532ull);
    for (size_t i = 0; i < count; ++i) out[i] = karney(&r, mean, stddev);
}

/* word (i & 7) of keystream block (i >> 3) of stream (s0, 0): the fixed-position draw of coefficient i
 * (mxx_amd/csrc/sampling.hip: one ChaCha20 block serves eight draws) */
static uint64_t positional_word(const uint64_t *seed, uint64_t s0, uint64_t s2, uint64_t tag, uint32_t i) {
    rng_t r;
    rng_init(&r, seed, s0, 0, s2, tag);
    r.state[12] = i >> 3;
    orc_chacha20_block(r.state, r.block);
    const uint32_t j = i & 7u;
    return (uint64_t)r.block[2 * j] | ((uint64_t)r.block[2 * j + 1] << 32);
}

/* Coefficient-domain samples of a rows x local_ncol window at column offset col_offset of a
 * rows x full_ncol matrix; out layout [poly][limb][n].  dist: 0 uniform, 1 gauss, 2 bit, 3 ternary.
 * Keying as in mxx_amd/csrc/sampling.hip (the reference's, cuda/src/matrix/MatrixSampling.cu:239-289, spends a
 * stream per residue): uniform = positional word of stream (gpoly + 1, 0) under sub-key (tag, limb + 1), rejected
 * words replaced by the first accepted word of the overflow stream (gpoly + 1, i + 1); bit / ternary = the
 * positional word under sub-key (tag, 0); Gaussian = Karney integers drawn one after the other from the stream of the
 * coefficient's pair, (gpoly + 1, (i >> 1) + 1). */
void orc_sample_distribution(uint64_t *out, size_t rows, size_t local_ncol, size_t full_ncol, size_t col_offset,
                             uint32_t L, uint32_t n, const uint64_t *moduli, int dist, double sigma,
                             const uint64_t *seed) {
    long total = (long)(rows * local_ncol);
#pragma omp parallel for schedule(static)
    for (long p = 0; p < total; ++p) {
        size_t row = (size_t)p / local_ncol, lcol = (size_t)p % local_ncol;
        uint64_t gpoly = row * full_ncol + col_offset + lcol;
        rng_t rg; /* Gaussian: one stream per pair of coefficients, drawn in order */
        memset(&rg, 0, sizeof(rg));
        for (uint32_t i = 0; i < n; ++i) {
            rng_t r;
            if (dist == 0) {
                for (uint32_t l = 0; l < L; ++l) {
                    const uint64_t q = moduli[l], max = ~0ull, threshold = max - (max % q);
                    uint64_t x = positional_word(seed, gpoly + 1, (uint64_t)l + 1, 0x6f70656e66686531ull, i);
                    if (x >= threshold) {
                        rng_init(&r, seed, gpoly + 1, (uint64_t)i + 1, (uint64_t)l + 1, 0x6f70656e66686531ull);
                        do x = rng_u64(&r); while (x >= threshold);
                    }
                    out[((size_t)p * L + l) * n + i] = x % q;
                }
                continue;
            }
            int64_t z;
            if (dist == 1) {
                if ((i & 1u) == 0) rng_init(&rg, seed, gpoly + 1, (uint64_t)(i >> 1) + 1, 0, 0x6f70656e66686532ull);
                z = karney(&rg, 0.0, sigma);
            } else if (dist == 2) {
                z = (int64_t)(positional_word(seed, gpoly + 1, 0, 0x6f70656e66686533ull, i) & 1ull);
            } else {
                uint64_t pick = positional_word(seed, gpoly + 1, 0, 0x6f70656e66686534ull, i) % 3ull;
                z = pick == 0 ? 0 : (pick == 1 ? 1 : -1);
            }
            for (uint32_t l = 0; l < L; ++l) out[((size_t)p * L + l) * n + i] = signed_mod(z, moduli[l]);
        }
    }
}

/* ---- the reference device RNG's OWN keying (MXX_HIP_RNG_COMPAT=reference) ----------------------------------------
 * Restates cuda/src/ChaCha.cu:104-167 and cuda/src/matrix/MatrixSampling.cu:6-147,239-289 as they stand, i.e. WITHOUT the
 * three departures documented in mxx_amd/csrc/rng.h: stream0 is the 64-bit block counter (state words 12, 13), stream1
 * the nonce (words 14, 15), a stream per (polynomial, coefficient[, limb]), 64-bit draws, 53-bit uniform deviates compared
 * as doubles.  With the switch set the library's gpu_matrix_sample_distribution(_columns) must produce these samples:
 * a matrix derived from a seed is then the one a CUDA build of the reference derives from it. */
typedef struct {
    uint32_t state[16], block[16];
    uint32_t idx; /* next 64-bit word of the block, 8 = empty */
} ref_rng_t;

static void ref_init(ref_rng_t *r, const uint64_t seed[4], uint64_t s0, uint64_t s1, uint64_t s2, uint64_t tag) {
    rng_t k;
    rng_init(&k, seed, 0, 0, s2, tag); /* the HChaCha20 sub-key is the same function of (seed, tag, stream2) */
    memcpy(r->state, k.state, 12 * sizeof(uint32_t));
    r->state[12] = (uint32_t)s0; r->state[13] = (uint32_t)(s0 >> 32); /* ChaCha.cu:138-149: the counter IS stream0 */
    r->state[14] = (uint32_t)s1; r->state[15] = (uint32_t)(s1 >> 32);
    r->idx = 8;
}

static uint64_t ref_u64(ref_rng_t *r) {
    if (r->idx >= 8) {
        orc_chacha20_block(r->state, r->block);
        if (++r->state[12] == 0) ++r->state[13];
        r->idx = 0;
    }
    const uint64_t v = (uint64_t)r->block[2 * r->idx] | ((uint64_t)r->block[2 * r->idx + 1] << 32);
    ++r->idx;
    return v;
}

static double ref_u01(ref_rng_t *r) {
    const double scale = 1.0 / 9007199254740992.0;
    double u = (double)(ref_u64(r) >> 11) * scale;
    if (u <= 0.0) u = scale;
    else if (u >= 1.0) u = 1.0 - scale;
    return u;
}

static int ref_h(ref_rng_t *r) { /* MatrixSampling.cu:30-50 */
    double a = ref_u01(r);
    if (!(a < 0.5)) return 1;
    for (;;) {
        const double b = ref_u01(r);
        if (!(b < a)) return 0;
        a = ref_u01(r);
        if (!(a < b)) return 1;
    }
}

static int ref_b(ref_rng_t *r, int32_t k, double x) { /* :74-96 */
    double y = x;
    int32_t n = 0;
    const double m = (double)(2 * k + 2);
    for (;; ++n) {
        const double z = ref_u01(r);
        if (!(z < y)) break;
        const double t = ref_u01(r);
        if (!(t < (2.0 * (double)k + x) / m)) break;
        y = z;
        if (n > 4096) break;
    }
    return (n % 2) == 0;
}

static int64_t ref_karney(ref_rng_t *r, double mean, double stddev) { /* :98-147 */
    if (!(stddev > 0.0) || !isfinite(mean) || !isfinite(stddev)) return (int64_t)llround(mean);
    const int64_t cs = (int64_t)ceil(stddev);
    if (cs <= 0) return (int64_t)llround(mean);
    for (int iter = 0; iter < (1 << 16); ++iter) {
        int32_t k = 0;
        while (ref_h(r)) {
            if (++k > 1024) break;
        }
        int32_t n = k * (k - 1);
        while (n-- && ref_h(r)) {
        }
        if (!(n < 0)) continue;
        const int64_t s = (ref_u64(r) & 1ull) ? 1 : -1;
        const double di0 = stddev * (double)k + (double)s * mean;
        const int64_t i0 = (int64_t)ceil(di0);
        const double x0 = ((double)i0 - di0) / stddev;
        const int64_t j = (int64_t)(ref_u64(r) % (uint64_t)cs);
        const double x = x0 + (double)j / stddev;
        if (!(x < 1.0) || (x == 0.0 && s < 0 && k == 0)) continue;
        int32_t h = k + 1;
        while (h-- > 0 && ref_b(r, k, x)) {
        }
        if (h >= 0) continue;
        return s * (i0 + j);
    }
    const double u1 = ref_u01(r), u2 = ref_u01(r); /* never reached in practice; libm here, as the reference */
    return (int64_t)llround(mean + stddev * (sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2)));
}

/* same layout as orc_sample_distribution; keys of MatrixSampling.cu:239-289 */
void orc_sample_distribution_refkey(uint64_t *out, size_t rows, size_t local_ncol, size_t full_ncol, size_t col_offset,
                                    uint32_t L, uint32_t n, const uint64_t *moduli, int dist, double sigma,
                                    const uint64_t *seed) {
    long total = (long)(rows * local_ncol);
#pragma omp parallel for schedule(static)
    for (long p = 0; p < total; ++p) {
        size_t row = (size_t)p / local_ncol, lcol = (size_t)p % local_ncol;
        uint64_t gpoly = row * full_ncol + col_offset + lcol;
        for (uint32_t i = 0; i < n; ++i) {
            ref_rng_t r;
            if (dist == 0) {
                for (uint32_t l = 0; l < L; ++l) {
                    const uint64_t q = moduli[l], max = ~0ull, threshold = max - (max % q);
                    ref_init(&r, seed, gpoly + 1, (uint64_t)i + 1, (uint64_t)l + 1, 0x6f70656e66686531ull);
                    uint64_t x;
                    do x = ref_u64(&r); while (x >= threshold);
                    out[((size_t)p * L + l) * n + i] = x % q;
                }
                continue;
            }
            int64_t z;
            ref_init(&r, seed, gpoly + 1, (uint64_t)i + 1, 0, 0x6f70656e66686531ull + (uint64_t)dist);
            if (dist == 1) z = ref_karney(&r, 0.0, sigma);
            else if (dist == 2) z = (int64_t)(ref_u64(&r) & 1ull);
            else {
                const uint64_t pick = ref_u64(&r) % 3ull;
                z = pick == 0 ? 0 : (pick == 1 ? 1 : -1);
            }
            for (uint32_t l = 0; l < L; ++l) out[((size_t)p * L + l) * n + i] = signed_mod(z, moduli[l]);
        }
    }
}

static inline uint32_t bit_width(uint64_t v) { return v ? 64 - (uint32_t)__builtin_clzll(v) : 0; }

/* G-lattice sampling, COEFF in (rows x cols) -> COEFF out (rows*k x cols), k = dpt*L. */
void orc_gauss_samp_gq(uint64_t *out, const uint64_t *src, size_t rows, size_t cols, uint32_t L, uint32_t n,
                       const uint64_t *moduli, uint32_t base_bits, double c, const uint64_t *seed) {
    uint32_t crt_bits = 0;
    for (uint32_t l = 0; l < L; ++l) if (bit_width(moduli[l]) > crt_bits) crt_bits = bit_width(moduli[l]);
    const uint32_t dpt = (crt_bits + base_bits - 1) / base_bits;
    const size_t k = (size_t)dpt * L;
    const uint64_t base = 1ull << base_bits;
    const double base_f = (double)base, sigma = c / (base_f + 1.0), kf = (double)dpt;
    long total = (long)(rows * cols);
#pragma omp parallel for schedule(static)
    for (long p = 0; p < total; ++p) {
        size_t row = (size_t)p / cols, col = (size_t)p % cols;
        int64_t md[64], vd[64], z[64];
        double l_[64], h_[64], cv[64], pp[64], a[64], zf[64];
        for (uint32_t t = 0; t < L; ++t) {
            uint64_t qt = moduli[t];
            uint64_t mq = qt;
            for (uint32_t d = 0; d < dpt; ++d) { md[d] = (int64_t)(mq % base); mq /= base; }
            l_[0] = sqrt(base_f * (1.0 + 1.0 / kf) + 1.0);
            for (uint32_t d = 1; d < dpt; ++d) l_[d] = sqrt(base_f * (1.0 + 1.0 / (kf - (double)d)));
            h_[0] = 0.0;
            for (uint32_t d = 1; d < dpt; ++d) h_[d] = sqrt(base_f * (1.0 - 1.0 / (kf - (double)(d - 1))));
            cv[0] = (double)md[0] / base_f;
            for (uint32_t d = 1; d < dpt; ++d) cv[d] = (cv[d - 1] + (double)md[d]) / base_f;
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t vv = src[((size_t)p * L + t) * n + i] % qt;
                for (uint32_t d = 0; d < dpt; ++d) { vd[d] = (int64_t)(vv % base); vv /= base; }
                rng_t r;
                /* one sub-key per call; stream0 = (coefficient + 1) 2^8 + (tower + 1), stream1 = polynomial + 1 */
                rng_init(&r, seed, (((uint64_t)i + 1) << 8) | ((uint64_t)t + 1), (uint64_t)p + 1, 0, 0x6761646765746731ull);
                for (uint32_t d = 0; d < dpt; ++d) zf[d] = sigma * std_normal(&r);
                for (uint32_t d = 0; d + 1 < dpt; ++d) pp[d] = l_[d] * zf[d] + h_[d + 1] * zf[d + 1];
                pp[dpt - 1] = h_[dpt - 1] * zf[dpt - 1];
                a[0] = ((double)vd[0] - pp[0]) / base_f;
                for (uint32_t d = 1; d < dpt; ++d) a[d] = (a[d - 1] + (double)vd[d] - pp[d]) / base_f;
                const uint32_t last = dpt - 1;
                z[last] = karney(&r, -a[last] / cv[last], sigma / cv[last]);
                for (uint32_t d = 0; d < dpt; ++d) a[d] += (double)z[last] * cv[d];
                for (uint32_t d = 0; d < last; ++d) z[d] = karney(&r, -a[d], sigma);
                for (uint32_t d = 0; d < dpt; ++d) {
                    int64_t digit;
                    if (dpt == 1) digit = (int64_t)base * z[0] + md[0] * z[0] + vd[0];
                    else if (d == 0) digit = (int64_t)base * z[0] + md[0] * z[last] + vd[0];
                    else if (d < last) digit = (int64_t)base * z[d] - z[d - 1] + md[d] * z[last] + vd[d];
                    else digit = md[last] * z[last] - z[last - 1] + vd[last];
                    size_t orow = row * k + (size_t)t * dpt + d;
                    for (uint32_t l = 0; l < L; ++l)
                        out[((orow * cols + col) * L + l) * n + i] = signed_mod(digit, moduli[l]);
                }
            }
        }
    }
}

/* p1 covariance factorisation from COEFF A,B,D (d x d, limb 0, centred): outputs
 * sqrt_var[n][m] and update[n][m][m] (m = 2d). */
void orc_p1_covariance(const uint64_t *a_mat, const uint64_t *b_mat, const uint64_t *d_mat, size_t d, uint32_t L,
                       uint32_t n, uint64_t q0, double sigma, double s, double dgg_stddev, double *sqrt_var,
                       double *update) {
    const size_t m = 2 * d;
    const double sigma2 = sigma * sigma, s2 = s * s, fallback = dgg_stddev * dgg_stddev, eps = 1e-9;
    double *cov = (double *)malloc(sizeof(double) * m * m);
    memset(update, 0, sizeof(double) * n * m * m);
    for (uint32_t i = 0; i < n; ++i) {
        for (size_t r = 0; r < d; ++r)
            for (size_t c = 0; c < d; ++c) {
                size_t rc = ((r * d + c) * L) * (size_t)n + i, cr = ((c * d + r) * L) * (size_t)n + i;
                double a_rc = (double)centered(a_mat[rc], q0), d_rc = (double)centered(d_mat[rc], q0);
                double b_rc = (double)centered(b_mat[rc], q0), b_cr = (double)centered(b_mat[cr], q0);
                cov[r * m + c] = -sigma2 * a_rc + (r == c ? s2 : 0.0);
                cov[(r + d) * m + (c + d)] = -sigma2 * d_rc + (r == c ? s2 : 0.0);
                cov[r * m + (c + d)] = -sigma2 * b_rc;
                cov[(r + d) * m + c] = -sigma2 * b_cr;
            }
        double *sv = sqrt_var + (size_t)i * m, *up = update + (size_t)i * m * m;
        for (int t = (int)m - 1; t >= 0; --t) {
            double var = cov[t * m + t];
            if (!(var > eps)) var = fallback;
            sv[t] = sqrt(var);
            for (int r = 0; r < t; ++r) up[t * m + r] = cov[r * m + t] / var;
            if (t == 0) break;
            for (int r = 0; r < t; ++r) {
                double cr_ = up[t * m + r];
                for (int c = 0; c <= r; ++c) {
                    double colc = up[t * m + c] * var;
                    double v = cov[r * m + c] - cr_ * colc;
                    cov[r * m + c] = v;
                    cov[c * m + r] = v;
                }
            }
        }
    }
    free(cov);
}

/* p1 sampling from the factorisation; tp2 COEFF (m x cols); out COEFF residues (m x cols). */
void orc_sample_p1(uint64_t *out, const uint64_t *tp2, size_t m, size_t cols, uint32_t L, uint32_t n,
                   const uint64_t *moduli, const double *sqrt_var, const double *update, double c_scale,
                   const uint64_t *seed) {
    double *mean = (double *)malloc(sizeof(double) * m);
    for (size_t col = 0; col < cols; ++col)
        for (uint32_t i = 0; i < n; ++i) {
            const double *sv = sqrt_var + (size_t)i * m, *up = update + (size_t)i * m * m;
            rng_t r;
            rng_init(&r, seed, (uint64_t)col + 1, (uint64_t)i + 1, 0, 0x7065727475726231ull);
            for (size_t row = 0; row < m; ++row)
                mean[row] = c_scale * (double)centered(tp2[((row * cols + col) * L) * (size_t)n + i], moduli[0]);
            for (int t = (int)m - 1; t >= 0; --t) {
                double mu = mean[t];
                int64_t z = karney(&r, mu, sv[t]);
                for (uint32_t l = 0; l < L; ++l)
                    out[(((size_t)t * cols + col) * L + l) * (size_t)n + i] = signed_mod(z, moduli[l]);
                double delta = (double)z - mu;
                for (int rr = 0; rr < t; ++rr) mean[rr] += up[t * m + rr] * delta;
            }
        }
    free(mean);
}
