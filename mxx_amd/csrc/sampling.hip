// sampling.hip — seeded distribution sampling straight into device matrices.
// Replaces cuda/src/matrix/MatrixSampling.cu behind
// cuda/include/matrix/MatrixSampling.cuh:25-37.
//
// Every sample is a pure function of (seed, GLOBAL polynomial index row*full_ncol + col,
// coefficient index and, for the uniform distribution, limb index):
// sample_distribution_columns therefore equals the matching column slice of the full
// sample for the same seed (reference: MatrixSampling.cu:232-289, test
// src/sampler/gpu.rs:323-361).  Gaussian: a ChaCha20 stream per pair of coefficients (Karney's
// sampler takes a variable number of 16-bit draws, rng.h).  Uniform / bit / ternary: fixed-position
// draws, eight per keystream block (the reference's keying, one stream - i.e. at least
// one block - per residue, made the generator the whole cost of the call).  Integers are
// drawn once and written as residues into every limb; the result is transformed to EVAL.
#include "common.h"
#include "modarith.h"
#include "rng.h"

#include <algorithm>

static constexpr uint64_t kTagUniform = 0x6f70656e66686531ull;
static constexpr uint64_t kTagGauss = 0x6f70656e66686532ull;
static constexpr uint64_t kTagBit = 0x6f70656e66686533ull;
static constexpr uint64_t kTagTernary = 0x6f70656e66686534ull;

// uniform / bit / ternary: ONE ChaCha20 block serves eight draws (round 2 spent a block per residue: 25 ms for M2A's
// 3.5 GB operand, 0.02 of the HBM roofline, with the generator as the whole cost).  Keying, a pure function of
// (seed, global polynomial index, limb, coefficient) as before, so column windows commute (src/sampler/gpu.rs:323-361):
//   uniform : sub-key = HChaCha20(seed, tag, stream2 = limb + 1); residue of coefficient i = word (i & 7) of keystream
//             block (i >> 3) of stream (gpoly + 1, 0), reduced mod q when it is below floor(2^64 / q) q; a rejected
//             word (probability q 2^-64) is replaced by the first accepted word of the coefficient's own overflow
//             stream (gpoly + 1, i + 1) under the same sub-key;
//   bit / ternary : sub-key (seed, tag, 0); the draw of coefficient i = the same word of stream (gpoly + 1, 0).
// A thread owns eight consecutive coefficients of one (polynomial, limb) vector: one block in registers, no LDS, and
// 32 / 64 contiguous bytes stored per lane.  The sub-keys are derived once per call by a one-wave launch.
__global__ void derive_subkeys_kernel(ChaChaKey *__restrict__ keys, GpuRngSeed seed, uint64_t tag, uint32_t count, uint32_t first_stream2) {
    const uint32_t l = threadIdx.x;
    if (l < count) keys[l] = chacha_subkey(seed, static_cast<uint64_t>(first_stream2) + l, tag);
}

__device__ __forceinline__ ChaChaKey load_key(const ChaChaKey *__restrict__ keys, uint32_t l) {
    const uint4 *p = reinterpret_cast<const uint4 *>(keys + l);
    const uint4 a = p[0], b = p[1];
    ChaChaKey k;
    k.w[0] = a.x; k.w[1] = a.y; k.w[2] = a.z; k.w[3] = a.w;
    k.w[4] = b.x; k.w[5] = b.y; k.w[6] = b.z; k.w[7] = b.w;
    return k;
}

// the rare path of the uniform sampler: first accepted word of the coefficient's overflow stream
__device__ __noinline__ uint64_t uniform_overflow_draw(const ChaChaKey &key, uint64_t stream0, uint64_t stream1, uint64_t threshold) {
    for (uint32_t block = 0;; ++block) {
        uint64_t w[8];
        chacha_block_words(key, stream0, stream1, block, w);
        for (int j = 0; j < 8; ++j)
            if (w[j] < threshold) return w[j];
    }
}

template <typename W, int VEC>
__device__ __forceinline__ void store_group(W *dst, const W (&r)[8], uint32_t count) {
    if (count == 8 && VEC) {
        typedef W vec_t __attribute__((ext_vector_type(16 / sizeof(W))));
        constexpr int per = 16 / sizeof(W);
#pragma unroll
        for (int v = 0; v < 8 / per; ++v) {
            vec_t x;
#pragma unroll
            for (int e = 0; e < per; ++e) x[e] = r[v * per + e];
            __builtin_nontemporal_store(x, reinterpret_cast<vec_t *>(dst) + v);
        }
    } else {
        for (uint32_t j = 0; j < count; ++j) dst[j] = r[j];
    }
}

template <typename W>
__global__ void __launch_bounds__(256) sample_uniform_kernel(W *__restrict__ out, const LimbConst *__restrict__ limbs,
                                                             const ChaChaKey *__restrict__ keys, size_t polys, size_t local_ncol,
                                                             size_t full_ncol, size_t col_offset, uint32_t L, uint32_t N,
                                                             uint32_t groups /* ceil(N / 8) */) {
    const size_t idx = item_index();
    if (idx >= polys * L * groups) return;
    const size_t vec = idx / groups;  // (polynomial, limb) vector
    const uint32_t g = static_cast<uint32_t>(idx - vec * groups);
    const size_t p = vec / L;
    const uint32_t l = static_cast<uint32_t>(vec - p * L);
    const size_t row = p / local_ncol, lcol = p - row * local_ncol;
    const uint64_t gpoly = row * full_ncol + col_offset + lcol;
    const ChaChaKey key = load_key(keys, l);
    const uint64_t q = limbs[l].q, mu64 = limbs[l].mu64;
    const uint64_t threshold = mu64 * q;  // = (2^64 - 1) - (2^64 - 1) % q: q is an odd prime, so floor((2^64-1)/q) = floor(2^64/q)
    uint64_t w[8];
    chacha_block_words(key, gpoly + 1, 0, g, w);
    const uint32_t count = min(8u, N - g * 8u);
    W r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint64_t x = w[j];
        if (x >= threshold && static_cast<uint32_t>(j) < count) x = uniform_overflow_draw(key, gpoly + 1, static_cast<uint64_t>(g) * 8u + j + 1, threshold);
        uint64_t rem = x - __umul64hi(x, mu64) * q;  // quotient estimate at most one short
        rem = rem >= q ? rem - q : rem;
        r[j] = static_cast<W>(rem);
    }
    store_group<W, 1>(out + vec * N + static_cast<size_t>(g) * 8u, r, count);
}

// bit / ternary: the draw is shared by all limbs
template <typename W>
__global__ void __launch_bounds__(256) sample_small_kernel(W *__restrict__ out, const LimbConst *__restrict__ limbs,
                                                           const ChaChaKey *__restrict__ keys, size_t polys, size_t local_ncol,
                                                           size_t full_ncol, size_t col_offset, uint32_t L, uint32_t N, uint32_t groups,
                                                           int dist) {
    const size_t idx = item_index();
    if (idx >= polys * groups) return;
    const size_t p = idx / groups;
    const uint32_t g = static_cast<uint32_t>(idx - p * groups);
    const size_t row = p / local_ncol, lcol = p - row * local_ncol;
    const uint64_t gpoly = row * full_ncol + col_offset + lcol;
    const ChaChaKey key = load_key(keys, 0);
    uint64_t w[8];
    chacha_block_words(key, gpoly + 1, 0, g, w);
    const uint32_t count = min(8u, N - g * 8u);
    int32_t z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (dist == GPU_MATRIX_DIST_BIT) {
            z[j] = static_cast<int32_t>(w[j] & 1ull);
        } else {
            const uint32_t pick = static_cast<uint32_t>(w[j] % 3ull);
            z[j] = pick == 0 ? 0 : (pick == 1 ? 1 : -1);
        }
    }
    for (uint32_t l = 0; l < L; ++l) {
        const W q = static_cast<W>(limbs[l].q);
        W r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = z[j] < 0 ? q - 1 : static_cast<W>(z[j]);
        store_group<W, 1>(out + (p * L + l) * N + static_cast<size_t>(g) * 8u, r, count);
    }
}

// discrete Gaussian: persistent lanes (rng.h).  An element is a PAIR of consecutive coefficients of one polynomial drawn
// one after the other from ONE stream (gpoly + 1, (i >> 1) + 1) - still a pure function of (seed, global index), column
// windows commute: the samples depend on no input.  Wave w owns pairs [w*64*per_lane, +64*per_lane); its lanes take
// them one at a time (wave_take).  Lanes finish at different times, so a lane's store is a lone 8 bytes: it goes to a
// compact int64 staging array ([poly][N]) that a coalesced pass expands into the L residues per coefficient (storing
// the residues from here cost a 32-byte HBM write per limb).
// Why pairs, and the refill policy.  With a stream per coefficient (~21 draws, ~20 steps, an empty ring at every start)
// about 25 of a wave's 64 lanes were out of draws at EVERY checkpoint and the block pass - counted: 96 % of the
// checkpoints, 1.74 blocks computed per coefficient for 0.66 consumed, ~70 % of the kernel's instructions - ran whatever
// the cadence.  A pair's stream lasts ~42 steps and wastes half as many draws, so refills can be held to every third
// checkpoint and are never forced (a lane that runs dry waits).  Same-box A/B of the group size (M3A call / M4 step):
// one coefficient per stream 7.60 ms / 0.634, pairs 7.13 / 0.647, quads 7.08 / 0.70 (a small launch lasts as long as
// its unluckiest lane's chain, which grows with the group), eight 7.2 / -.  Pairs.
#ifndef GAUSS_GROUP_LOG
#define GAUSS_GROUP_LOG 1  // two coefficients per stream (the CPU restatement's keying: oracle/oracle_sampling.c)
#endif
// UNI: every wave's chunk of groups lies inside one polynomial (the launcher checks that 64 * per_lane divides the groups
// per polynomial): its (row, column) - a 32-bit division - is then computed once per wave on the scalar unit instead of in
// every group hand-over
// SEG (with UNI): the matrix is the column-wise concatenation of independently seeded segments (rng.h, RngSegments); the
// wave's polynomial lies in one segment, whose sub-key and shape replace `key`, `full_ncol` and `col_offset` - all
// wave-uniform, read from the kernel argument on the scalar unit.
template <int SV, bool UNI = false, bool SEG = false>
__global__ void __launch_bounds__(SAMPLER_THREADS) sample_gauss_kernel(int64_t *__restrict__ stage, size_t polys,
                                    uint32_t local_ncol, size_t full_ncol, size_t col_offset, uint32_t logN,
                                    double sigma, KarneyDivisor div, ChaChaKey key, uint32_t per_lane, uint32_t fill_every,
                                    int starve_limit, SegArg<SEG> segs = SegArg<SEG>{}) {
    static_assert(!SEG || UNI, "segments need a wave's chunk inside one polynomial");
    __shared__ uint32_t ring[SAMPLER_THREADS * RNG_RING_SLOTS];  // 128 bytes per lane: 8 KB per one-wave workgroup
    const uint32_t glog = logN < GAUSS_GROUP_LOG ? logN : GAUSS_GROUP_LOG;
    const uint32_t G = 1u << glog;
    const size_t total = (polys << logN) >> glog;  // groups
    WaveChunk chunk = wave_chunk(total, per_lane);
    const uint32_t p_u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(chunk.base >> (logN - glog))));
    const uint32_t row_u = p_u / local_ncol, lcol_u = p_u - row_u * local_ncol;
    uint64_t stream0_u = row_u * full_ncol + col_offset + lcol_u + 1;
    if constexpr (SEG) {
        const uint32_t j = rng_segment_of(segs, lcol_u);
        key = segs.key[j];
        stream0_u = static_cast<uint64_t>(row_u) * (segs.start[j + 1] - segs.start[j]) + (lcol_u - segs.start[j]) + 1;
    }
    ChaChaRng rng;
    rng_init_keyed(rng, ring, key, 0, 0, SAMPLER_THREADS);
    KarneyFsm f;
    karney_reset(f);
    karney_fix_parameters(f, 0.0, sigma, div);  // one centre and one width for the whole matrix
    const bool degenerate = div.degenerate;
    bool have = false, fin = true;
    uint32_t cnt = 0;
    size_t idx = 0;
    auto integer_ready = [&]() {
        if (f.st == KS_DONE && !fin) {
            stage[(idx << glog) + cnt] = f.result;
            ++cnt;
            if (cnt == G) fin = true;
            else karney_begin_fixed(f, degenerate);
        }
    };
    for (uint32_t step = 0;; ++step) {  // one superstep per iteration
        integer_ready();
        {
            const bool take = f.st == KS_DONE && fin;
            const uint32_t e = wave_take(chunk, take);
            if (take) {
                have = e < chunk.len;
                if (have) {
                    idx = chunk.base + e;
                    uint64_t stream0;
                    if constexpr (UNI) {
                        stream0 = stream0_u;
                    } else {
                        const uint32_t p = static_cast<uint32_t>(idx >> (logN - glog));
                        const uint32_t row = p / local_ncol, lcol = p - row * local_ncol;
                        stream0 = row * full_ncol + col_offset + lcol + 1;
                    }
                    rng_reopen(rng, stream0, (idx & ((size_t(1) << (logN - glog)) - 1)) + 1);
                    cnt = 0;
                    fin = false;
                    karney_begin_fixed(f, degenerate);
                } else {
                    f.st = KS_IDLE;
                }
            }
            if (__all(f.st == KS_IDLE)) break;
            rng_fill_wave(rng, f.st != KS_IDLE, step % fill_every == 0, starve_limit, karney_urgent(SV));
        }
        karney_heavy(f, rng);
#pragma unroll
        for (int s = 0; s < KARNEY_SUPERSTEP / SV; ++s) karney_light(f, rng);
#pragma unroll
        for (int sv = 1; sv < SV; ++sv) {
            integer_ready();
            karney_heavy(f, rng);
#pragma unroll
            for (int s = 0; s < KARNEY_SUPERSTEP / SV; ++s) karney_light(f, rng);
        }
    }
}

// ---- MXX_HIP_RNG_COMPAT=reference: the reference device RNG's own keying -----------------------
// The default keying departs from cuda/src/ChaCha.cu:138-149 + cuda/src/matrix/MatrixSampling.cu:239-289 in three
// documented ways (rng.h), so a matrix derived from a seed - the hash sampler's included - differs from the one a CUDA
// build derives.  With the switch set (read per context at creation, gpupoly_reload_env) gpu_matrix_sample_distribution
// and _columns key and consume their streams exactly as the reference does: a stream per (polynomial, coefficient) -
// and limb, for the uniform distribution -, stream0 in the 64-bit block counter, stream1 in the nonce, 64-bit draws, Karney
// on 53-bit deviates compared as doubles.  One thread per coefficient, nested loops: the slow path, for interoperability
// only.  Pure function of (seed, global polynomial index, coefficient, limb) as before: column windows commute.
struct RefRng {
    uint32_t state[16], block[16];
    uint32_t idx;
};

__device__ __forceinline__ void ref_rng_init(RefRng &r, const GpuRngSeed &seed, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t tag) {
    const ChaChaKey k = chacha_subkey(seed, s2, tag);
    r.state[0] = 0x61707865u; r.state[1] = 0x3320646eu; r.state[2] = 0x79622d32u; r.state[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.state[4 + i] = k.w[i];
    r.state[12] = static_cast<uint32_t>(s0);
    r.state[13] = static_cast<uint32_t>(s0 >> 32);
    r.state[14] = static_cast<uint32_t>(s1);
    r.state[15] = static_cast<uint32_t>(s1 >> 32);
    r.idx = 8;
}

__device__ __noinline__ void ref_rng_refill(RefRng &r) {
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = r.state[i];
    chacha_rounds(x);
    for (int i = 0; i < 16; ++i) r.block[i] = x[i] + r.state[i];
    if (++r.state[12] == 0) ++r.state[13];
    r.idx = 0;
}

__device__ __forceinline__ uint64_t ref_rng_u64(RefRng &r) {
    if (r.idx >= 8) ref_rng_refill(r);
    uint32_t lo = 0, hi = 0;
    for (uint32_t i = 0; i < 8; ++i)  // register-resident block: select instead of indexing
        if (i == r.idx) { lo = r.block[2 * i]; hi = r.block[2 * i + 1]; }
    ++r.idx;
    return static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
}

__device__ __forceinline__ double ref_rng_u01(RefRng &r) { return u64_to_open01(ref_rng_u64(r)); }

__device__ bool ref_karney_h(RefRng &r) {
    double a = ref_rng_u01(r);
    if (!(a < 0.5)) return true;
    for (;;) {
        const double b = ref_rng_u01(r);
        if (!(b < a)) return false;
        a = ref_rng_u01(r);
        if (!(a < b)) return true;
    }
}

__device__ bool ref_karney_b(RefRng &r, int32_t k, double x) {
    double y = x;
    int32_t n = 0;
    const double m = static_cast<double>(2 * k + 2);
    for (;; ++n) {
        const double z = ref_rng_u01(r);
        if (!(z < y)) break;
        const double t = ref_rng_u01(r);
        if (!(t < (2.0 * static_cast<double>(k) + x) / m)) break;
        y = z;
        if (n > 4096) break;
    }
    return (n % 2) == 0;
}

__device__ int64_t ref_karney(RefRng &r, double mean, double stddev) {
    if (!(stddev > 0.0) || !isfinite(mean) || !isfinite(stddev)) return static_cast<int64_t>(llround(mean));
    const int64_t cs = static_cast<int64_t>(ceil(stddev));
    if (cs <= 0) return static_cast<int64_t>(llround(mean));
    for (int iter = 0; iter < (1 << 16); ++iter) {
        int32_t k = 0;
        while (ref_karney_h(r)) {
            if (++k > 1024) break;
        }
        int32_t n = k * (k - 1);
        while (n-- && ref_karney_h(r)) {
        }
        if (!(n < 0)) continue;
        const int64_t s = (ref_rng_u64(r) & 1ull) ? 1 : -1;
        const double di0 = stddev * static_cast<double>(k) + static_cast<double>(s) * mean;
        const int64_t i0 = static_cast<int64_t>(ceil(di0));
        const double x0 = (static_cast<double>(i0) - di0) / stddev;
        const int64_t j = static_cast<int64_t>(ref_rng_u64(r) % static_cast<uint64_t>(cs));
        const double x = x0 + static_cast<double>(j) / stddev;
        if (!(x < 1.0) || (x == 0.0 && s < 0 && k == 0)) continue;
        int32_t h = k + 1;
        while (h-- > 0 && ref_karney_b(r, k, x)) {
        }
        if (h >= 0) continue;
        return s * (i0 + j);
    }
    // 2^16 rejected trials: not reachable in practice (the reference falls back to a rounded normal here)
    const double u1 = ref_rng_u01(r), u2 = ref_rng_u01(r);
    return static_cast<int64_t>(llround(mean + stddev * (sqrt(-2.0 * det_log(u1)) * det_cos2pi(u2))));
}

template <typename W>
__global__ void __launch_bounds__(128) sample_compat_kernel(W *__restrict__ out, const LimbConst *__restrict__ limbs, size_t polys,
                                                            size_t local_ncol, size_t full_ncol, size_t col_offset, uint32_t L,
                                                            uint32_t logN, int dist, double sigma, GpuRngSeed seed) {
    const size_t idx = item_index();
    if (idx >= (polys << logN)) return;
    const size_t p = idx >> logN;
    const uint32_t i = static_cast<uint32_t>(idx & ((size_t(1) << logN) - 1));
    const size_t row = p / local_ncol, lcol = p - row * local_ncol;
    const uint64_t gpoly = row * full_ncol + col_offset + lcol;
    RefRng r;
    if (dist == GPU_MATRIX_DIST_UNIFORM) {
        for (uint32_t l = 0; l < L; ++l) {
            const uint64_t q = limbs[l].q, threshold = ~0ull - (~0ull % q);
            ref_rng_init(r, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, static_cast<uint64_t>(l) + 1, kTagUniform);
            uint64_t x;
            do x = ref_rng_u64(r); while (x >= threshold);
            out[((p * L + l) << logN) + i] = static_cast<W>(x % q);
        }
        return;
    }
    int64_t z;
    if (dist == GPU_MATRIX_DIST_GAUSS) {
        ref_rng_init(r, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, 0, kTagGauss);
        z = ref_karney(r, 0.0, sigma);
    } else if (dist == GPU_MATRIX_DIST_BIT) {
        ref_rng_init(r, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, 0, kTagBit);
        z = static_cast<int64_t>(ref_rng_u64(r) & 1ull);
    } else {
        ref_rng_init(r, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, 0, kTagTernary);
        const uint64_t pick = ref_rng_u64(r) % 3ull;
        z = pick == 0 ? 0 : (pick == 1 ? 1 : -1);
    }
    for (uint32_t l = 0; l < L; ++l)
        out[((p * L + l) << logN) + i] = signed_to_residue_mu<W>(z, limbs[l].q, limbs[l].mu64);
}

// keep_coeff: leave the samples as coefficients (for a caller that decomposes them next) instead of finishing in EVAL
int sample_impl(GpuMatrix *out, int dist, double sigma, GpuRngSeed seed, size_t full_ncol, size_t col_offset, bool keep_coeff) {
    if (!out) return set_error("gpu_matrix_sample_distribution: null matrix");
    if (dist < GPU_MATRIX_DIST_UNIFORM || dist > GPU_MATRIX_DIST_TERNARY)
        return set_error("gpu_matrix_sample_distribution: invalid dist_type");
    if (dist == GPU_MATRIX_DIST_GAUSS && !(sigma > 0.0))
        return set_error("gpu_matrix_sample_distribution: sigma must be positive for Gaussian sampling");
    if (col_offset + out->cols > full_ncol)
        return set_error("gpu_matrix_sample_distribution_columns: column window out of range");
    // stream ids (global polynomial index + 1) are 48 bits wide in the nonce layout of rng.h: refuse what would wrap
    if (full_ncol && (out->rows > ((size_t(1) << 48) - 2) / full_ncol))
        return set_error("gpu_matrix_sample_distribution: matrix too large for the RNG's 48-bit stream ids");
    GpuContext *ctx = out->ctx;
    out->format = keep_coeff ? GPU_POLY_FORMAT_COEFF : GPU_POLY_FORMAT_EVAL;
    const size_t polys = matrix_polys(out);
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    const size_t total = polys * static_cast<size_t>(ctx->N);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out));
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    if (ctx->env.rng_compat) {
        const dim3 blocks = item_grid(total, 128);
        MXX_TRACE_BYTES(static_cast<double>(out->bytes));
        if (ctx->wide)
            MXX_LAUNCH(sample_compat_kernel<uint64_t>, blocks, dim3(128), 0, ctx->stream, static_cast<uint64_t *>(out->data), ctx->d_limbs,
                       polys, out->cols, full_ncol, col_offset, L, ctx->logN, dist, sigma, seed);
        else
            MXX_LAUNCH(sample_compat_kernel<uint32_t>, blocks, dim3(128), 0, ctx->stream, static_cast<uint32_t *>(out->data), ctx->d_limbs,
                       polys, out->cols, full_ncol, col_offset, L, ctx->logN, dist, sigma, seed);
        HIP_TRY(hipGetLastError());
        if (keep_coeff) return 0;
        return launch_ntt(ctx, out->data, polys * L, static_cast<int>(L), false);
    }
    if (dist == GPU_MATRIX_DIST_GAUSS) {
        // enough lanes to fill the chip first, then up to 16 coefficients per lane
        if (polys >> 32) return set_error("gpu_matrix_sample_distribution: too many polynomials");
        const size_t groups = total >> (ctx->logN < GAUSS_GROUP_LOG ? ctx->logN : GAUSS_GROUP_LOG);
        const uint32_t per_lane = sampler_per_lane(groups, reinterpret_cast<const void *>(sample_gauss_kernel<KARNEY_SERVICES>), ctx->device, ctx->env.sampler_per_lane);
        const unsigned blocks = static_cast<unsigned>((groups + SAMPLER_THREADS * per_lane - 1) / (SAMPLER_THREADS * per_lane));
        const KarneyDivisor div = karney_divisor(sigma);
        const ChaChaKey key = chacha_subkey(seed, 0, kTagGauss);
        void *stage = nullptr;
        if (ctx_alloc(ctx, total * sizeof(int64_t), &stage)) return 1;
        MXX_TRACE_BYTES(static_cast<double>(total) * sizeof(int64_t));  // no input: the int64 samples written once
        // one group per lane at most: the call lasts as long as its unluckiest lane's chain, which must not wait for
        // keystream - refills at every checkpoint, one service point per superstep; otherwise every third checkpoint and
        // never forced (64 lanes cannot reach 65)
        const size_t groups_per_poly = static_cast<size_t>(ctx->N) >> (ctx->logN < GAUSS_GROUP_LOG ? ctx->logN : GAUSS_GROUP_LOG);
        const bool uni = groups_per_poly % (static_cast<size_t>(SAMPLER_THREADS) * per_lane) == 0;
#define LAUNCH_SG(SV, UNI)                                                                                                     \
    MXX_LAUNCH((sample_gauss_kernel<SV, UNI>), dim3(blocks), dim3(SAMPLER_THREADS), 0, ctx->stream, static_cast<int64_t *>(stage), polys, \
               static_cast<uint32_t>(out->cols), full_ncol, col_offset, ctx->logN, sigma, div, key, per_lane,                   \
               static_cast<uint32_t>(ctx->env.sampler_fill_every ? ctx->env.sampler_fill_every : (per_lane == 1 ? 1 : 3)),      \
               per_lane == 1 ? 1 : 65, NoSegments{})
        if (per_lane == 1) {
            if (uni) LAUNCH_SG(1, true);
            else LAUNCH_SG(1, false);
        } else {
            if (uni) LAUNCH_SG(KARNEY_SERVICES, true);
            else LAUNCH_SG(KARNEY_SERVICES, false);
        }
#undef LAUNCH_SG
        const hipError_t err = hipGetLastError();
        const int rc = err == hipSuccess ? launch_scatter_i64(out, static_cast<const int64_t *>(stage)) : 0;
        ctx_free(ctx, stage);
        HIP_TRY(err);
        if (rc) return rc;
    } else {
        const bool uniform = dist == GPU_MATRIX_DIST_UNIFORM;
        const uint32_t groups = (N + 7) / 8;
        const uint32_t nkeys = uniform ? L : 1u;
        CtxBlock keys(ctx);  // per call: concurrent callers on one context must not share a key table
        if (keys.alloc(sizeof(ChaChaKey) * nkeys)) return 1;
        ChaChaKey *d_keys = static_cast<ChaChaKey *>(keys.ptr);
        const uint64_t tag = uniform ? kTagUniform : (dist == GPU_MATRIX_DIST_BIT ? kTagBit : kTagTernary);
        MXX_LAUNCH(derive_subkeys_kernel, dim3(1), dim3(64), 0, ctx->stream, d_keys, seed, tag, nkeys, uniform ? 1u : 0u);
        HIP_TRY(hipGetLastError());
        const size_t threads = polys * (uniform ? L : 1u) * groups;
        const dim3 blocks = item_grid(threads, 256);
#define MXX_SAMPLE(KERNEL, WORD, ...)                                                                                     \
    MXX_LAUNCH(KERNEL<WORD>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<WORD *>(out->data), ctx->d_limbs, \
                       d_keys, polys, out->cols, full_ncol, col_offset, L, N, groups, ##__VA_ARGS__)
        MXX_TRACE_BYTES(static_cast<double>(out->bytes));  // no input: the residues written once
        if (uniform) {
            if (ctx->wide) MXX_SAMPLE(sample_uniform_kernel, uint64_t);
            else MXX_SAMPLE(sample_uniform_kernel, uint32_t);
        } else {
            if (ctx->wide) MXX_SAMPLE(sample_small_kernel, uint64_t, dist);
            else MXX_SAMPLE(sample_small_kernel, uint32_t, dist);
        }
#undef MXX_SAMPLE
    }
    HIP_TRY(hipGetLastError());
    if (keep_coeff) return 0;
    // samples are coefficients; callers always get EVAL (MatrixSampling.cu:463-469)
    return launch_ntt(ctx, out->data, polys * L, static_cast<int>(L), false);
}

// ---- several independently seeded requests in one launch (rng.h, RngSegments) ---------------------------------------
// out = [S_0 | S_1 | ...], S_j = what gpu_matrix_sample_distribution writes for a rows x seg_cols[j] matrix under seeds[j].
// Gaussian only (the perturbation p2 of batched preimage requests); finishes in EVAL like the plain entry point.
extern "C" int gpupoly_matrix_sample_distribution_segments(GpuMatrix *out, int dist_type, double sigma, const GpuRngSeed *seeds,
                                                           const size_t *seg_cols, size_t nseg) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpupoly_matrix_sample_distribution_segments: null matrix");
    if (dist_type != GPU_MATRIX_DIST_GAUSS)
        return set_error("gpupoly_matrix_sample_distribution_segments: unsupported: only the Gaussian distribution is segmented");
    if (!(sigma > 0.0)) return set_error("gpupoly_matrix_sample_distribution_segments: sigma must be positive for Gaussian sampling");
    GpuContext *ctx = out->ctx;
    if (ctx->env.rng_compat)
        return set_error("gpupoly_matrix_sample_distribution_segments: unsupported under MXX_HIP_RNG_COMPAT=reference");
    RngSegments segs;
    if (!rng_segments_build(segs, seeds, seg_cols, nseg, kTagGauss, out->cols))
        return set_error("gpupoly_matrix_sample_distribution_segments: segments must be 1..64, non-empty and cover the matrix's columns");
    const size_t polys = matrix_polys(out);
    const uint32_t glog = ctx->logN < GAUSS_GROUP_LOG ? ctx->logN : GAUSS_GROUP_LOG;
    const size_t groups_per_poly = static_cast<size_t>(ctx->N) >> glog;
    if (groups_per_poly % SAMPLER_THREADS)
        return set_error("gpupoly_matrix_sample_distribution_segments: unsupported: ring too small for a wave per polynomial chunk");
    if (polys >> 32) return set_error("gpupoly_matrix_sample_distribution_segments: too many polynomials");
    out->format = GPU_POLY_FORMAT_EVAL;
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    const size_t total = polys * static_cast<size_t>(ctx->N);
    const size_t groups = total >> glog;
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out));
    // the plain launcher's lane count, cut down to a divisor of a polynomial's groups: a wave's chunk stays inside one
    // polynomial, i.e. inside one segment
    uint32_t per_lane = sampler_per_lane(groups, reinterpret_cast<const void *>(sample_gauss_kernel<KARNEY_SERVICES, true, true>), ctx->device, ctx->env.sampler_per_lane);
    while ((groups_per_poly / SAMPLER_THREADS) % per_lane) --per_lane;
    const unsigned blocks = static_cast<unsigned>((groups + SAMPLER_THREADS * per_lane - 1) / (SAMPLER_THREADS * per_lane));
    const KarneyDivisor div = karney_divisor(sigma);
    void *stage = nullptr;
    if (ctx_alloc(ctx, total * sizeof(int64_t), &stage)) return 1;
    MXX_TRACE_BYTES(static_cast<double>(total) * sizeof(int64_t));
#define LAUNCH_SGS(SV)                                                                                                         \
    MXX_LAUNCH((sample_gauss_kernel<SV, true, true>), dim3(blocks), dim3(SAMPLER_THREADS), 0, ctx->stream, static_cast<int64_t *>(stage), polys, \
               static_cast<uint32_t>(out->cols), out->cols, size_t(0), ctx->logN, sigma, div, ChaChaKey{}, per_lane,            \
               static_cast<uint32_t>(ctx->env.sampler_fill_every ? ctx->env.sampler_fill_every : (per_lane == 1 ? 1 : 3)),      \
               per_lane == 1 ? 1 : 65, segs)
    if (per_lane == 1) LAUNCH_SGS(1);
    else LAUNCH_SGS(KARNEY_SERVICES);
#undef LAUNCH_SGS
    const hipError_t err = hipGetLastError();
    const int rc = err == hipSuccess ? launch_scatter_i64(out, static_cast<const int64_t *>(stage)) : 0;
    ctx_free(ctx, stage);
    HIP_TRY(err);
    if (rc) return rc;
    return launch_ntt(ctx, out->data, polys * L, static_cast<int>(L), false);
    ABI_GUARD_END
}

// ---- detmath.h where it runs (extension, test instrument) ------------------------------------------------------
// Box-Muller's log / cos(2 pi u) are fixed IEEE operation sequences compiled into the device code; this entry evaluates
// them ON THE DEVICE for caller-supplied arguments so that a test can compare them with libm / extended precision
// (tests/test_gpu_sampler_stats.py: 10^7 points, <= 2 ulp; cos(2 pi x) <= 3 ulp) - the CPU-side check alone compares the header with itself.
__global__ void detmath_eval_kernel(double *__restrict__ out, const double *__restrict__ in, size_t n, int fn) {
    const size_t i = item_index();
    if (i >= n) return;
    const double x = in[i];
    out[i] = fn == 0 ? det_log(x) : (fn == 1 ? det_cos2pi(x) : sqrt(-2.0 * det_log(x)));
}

extern "C" int gpupoly_detmath_eval(GpuContext *ctx, int fn, const double *host_in, double *host_out, size_t n) {
    ABI_GUARD_BEGIN
    if (!ctx || !host_in || !host_out) return set_error("gpupoly_detmath_eval: null argument");
    if (fn < 0 || fn > 2) return set_error("gpupoly_detmath_eval: fn must be 0 (log), 1 (cos 2 pi u) or 2 (sqrt(-2 log u))");
    if (n == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    CtxBlock din(ctx), dout(ctx);
    if (din.alloc(n * sizeof(double)) || dout.alloc(n * sizeof(double))) return 1;
    HIP_TRY(hipMemcpyAsync(din.ptr, host_in, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MXX_LAUNCH(detmath_eval_kernel, item_grid(n, 256), dim3(256), 0, ctx->stream, static_cast<double *>(dout.ptr),
               static_cast<const double *>(din.ptr), n, fn);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_out, dout.ptr, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_sample_distribution(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpu_matrix_sample_distribution: null matrix");
    return sample_impl(out, dist_type, sigma, seed, out->cols, 0, false);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_sample_distribution_columns(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed,
                                                      size_t full_ncol, size_t col_offset) {
    ABI_GUARD_BEGIN
    return sample_impl(out, dist_type, sigma, seed, full_ncol, col_offset, false);
    ABI_GUARD_END
}
