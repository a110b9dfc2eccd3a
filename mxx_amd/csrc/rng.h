// rng.h — counter-based ChaCha20 streams and the integer samplers built on them.
//
// Stream keying follows the reference's device RNG so that a sample is a pure
// function of (seed, stream words, domain tag) and column windows commute
// (cuda/src/ChaCha.cu:104-167; keys in cuda/src/matrix/MatrixSampling.cu:239-289,
// cuda/src/matrix/MatrixTrapdoor.cu:234-241,772-779):
//   subkey  = HChaCha20(key = seed words (LE), nonce = domain_tag || stream2)
//   state   = "expand 32-byte k" | subkey | counter32 = 0 | nonce96 = (stream0, stream1)
//   output  = successive ChaCha20 blocks (32-bit block counter), read as 8 LE u64 words each.
// One departure from the reference, deliberate: ChaCha.cu:138-149 puts stream0 into the block
// counter words, so block b of stream (s0, s1) is block 0 of stream (s0 + b, s1) - and stream ids
// are consecutive everywhere (poly+1, column+1, tower+1), i.e. neighbouring polynomials, columns
// and towers would share keystream one block apart.  Here the counter is a pure block counter and
// the stream words live in the 96-bit nonce (RFC 8439 layout): stream0 in words 13 and 15[0:16],
// stream1 in words 14 and 15[16:32] (48 bits each; every stream id on this path is a polynomial,
// column, tower or coefficient index + 1).  No reference test pins keystream bytes; the
// properties callers rely on (pure function of (seed, global index), column windows commute,
// src/sampler/gpu.rs:292-361) are unchanged.
// Discrete Gaussians use Karney's exact rejection sampler with the reference's
// iteration caps (cuda/src/matrix/MatrixSampling.cu:30-147): only IEEE double
// compare / add / mul / div / ceil are involved, so the CPU oracle reproduces the
// stream bit for bit (both sides are built with -ffp-contract=off).
//
// Wave64 shape of the sampler.  A rejection sampler written as nested data-dependent
// loops makes a wave pay the product of the per-level maxima over its 64 lanes, and a
// ChaCha block refill inside such loops is executed once per straggling lane.  Here
//   * every lane owns a 16-word ring of keystream in LDS; blocks are generated only at
//     explicit, wave-convergent checkpoints (rng_fill), one per 8 draws at most;
//   * Karney's algorithm is a flat state machine that consumes exactly one keystream
//     word per step, so all lanes of a wave sit at the same program point and a wave
//     pays max-over-lanes of the TOTAL draw count once.
// The sequence of draws and decisions per lane is unchanged (bit-identical samples).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "detmath.h"

struct ChaChaRng {
    uint32_t state[16];
    uint64_t *ring;    // this lane's ring: element s lives at ring[s * ring_stride]
    uint32_t ring_stride;
    uint32_t head;     // words consumed
    uint32_t tail;     // words generated
};

// LDS bytes a block of `threads` lanes needs for its rings
#define RNG_RING_WORDS 16
#define RNG_LDS_BYTES(threads) ((threads) * RNG_RING_WORDS * sizeof(uint64_t))

__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

// UNROLL = 10 removes the register shuffling a rolled loop needs inside large kernels (about a
// third of the block cost there); the cold uses keep the rolled loop
template <int UNROLL = 1>
__host__ __device__ __forceinline__ void chacha_rounds(uint32_t (&x)[16]) {
#pragma unroll UNROLL
    for (int i = 0; i < 10; ++i) {
        CHACHA_QR(x[0], x[4], x[8], x[12])
        CHACHA_QR(x[1], x[5], x[9], x[13])
        CHACHA_QR(x[2], x[6], x[10], x[14])
        CHACHA_QR(x[3], x[7], x[11], x[15])
        CHACHA_QR(x[0], x[5], x[10], x[15])
        CHACHA_QR(x[1], x[6], x[11], x[12])
        CHACHA_QR(x[2], x[7], x[8], x[13])
        CHACHA_QR(x[3], x[4], x[9], x[14])
    }
}

// HChaCha20 sub-key of a stream family: depends on (seed, domain tag, stream2) only, so a kernel
// whose stream2 is constant (or takes few values) derives it once instead of once per coefficient
struct ChaChaKey {
    uint32_t w[8];
};

__host__ __device__ __forceinline__ ChaChaKey chacha_subkey(const GpuRngSeed &seed, uint64_t stream2, uint64_t domain_tag) {
    uint32_t x[16];
    x[0] = 0x61707865u; x[1] = 0x3320646eu; x[2] = 0x79622d32u; x[3] = 0x6b206574u;
    for (int i = 0; i < 4; ++i) {
        x[4 + 2 * i] = static_cast<uint32_t>(seed.words[i]);
        x[5 + 2 * i] = static_cast<uint32_t>(seed.words[i] >> 32);
    }
    x[12] = static_cast<uint32_t>(domain_tag);
    x[13] = static_cast<uint32_t>(domain_tag >> 32);
    x[14] = static_cast<uint32_t>(stream2);
    x[15] = static_cast<uint32_t>(stream2 >> 32);
    chacha_rounds(x);  // HChaCha20: no feed-forward, subkey = words 0..3 and 12..15
    ChaChaKey k;
    k.w[0] = x[0]; k.w[1] = x[1]; k.w[2] = x[2]; k.w[3] = x[3];
    k.w[4] = x[12]; k.w[5] = x[13]; k.w[6] = x[14]; k.w[7] = x[15];
    return k;
}

// ring_base: the block's LDS array of blockDim.x * 16 u64; call from every lane
// counter / nonce words of stream (stream0, stream1), positioned at keystream block `block`
__host__ __device__ __forceinline__ void chacha_set_stream(uint32_t (&state)[16], uint64_t stream0, uint64_t stream1,
                                                           uint32_t block) {
    state[12] = block;
    state[13] = static_cast<uint32_t>(stream0);
    state[14] = static_cast<uint32_t>(stream1);
    state[15] = (static_cast<uint32_t>(stream0 >> 32) & 0xffffu) | (static_cast<uint32_t>(stream1 >> 32) << 16);
}

// One keystream block of stream (stream0, stream1) under `key`, entirely in registers: 8 little-endian u64 words.
// The block-per-8-draws samplers (uniform, bit, ternary) use it: no ring, no LDS.
__host__ __device__ __forceinline__ void chacha_block_words(const ChaChaKey &key, uint64_t stream0, uint64_t stream1, uint32_t block,
                                                            uint64_t (&out)[8]) {
    uint32_t st[16], x[16];
    st[0] = 0x61707865u; st[1] = 0x3320646eu; st[2] = 0x79622d32u; st[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) st[4 + i] = key.w[i];
    chacha_set_stream(st, stream0, stream1, block);
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = st[i];
    chacha_rounds<10>(x);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        out[i] = static_cast<uint64_t>(x[2 * i] + st[2 * i]) | (static_cast<uint64_t>(x[2 * i + 1] + st[2 * i + 1]) << 32);
}

__device__ __forceinline__ void rng_init_keyed(ChaChaRng &rng, uint64_t *ring_base, const ChaChaKey &key,
                                               uint64_t stream0, uint64_t stream1) {
    rng.state[0] = 0x61707865u; rng.state[1] = 0x3320646eu; rng.state[2] = 0x79622d32u; rng.state[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) rng.state[4 + i] = key.w[i];
    chacha_set_stream(rng.state, stream0, stream1, 0);
    rng.ring = ring_base + threadIdx.x;
    rng.ring_stride = blockDim.x;
    rng.head = 0;
    rng.tail = 0;
}

__device__ __forceinline__ void rng_init(ChaChaRng &rng, uint64_t *ring_base, const GpuRngSeed &seed, uint64_t stream0,
                                         uint64_t stream1, uint64_t stream2, uint64_t domain_tag) {
    rng_init_keyed(rng, ring_base, chacha_subkey(seed, stream2, domain_tag), stream0, stream1);
}

// Re-key an open generator to another stream of the same family, positioned at keystream block
// `block`; buffered words are dropped
__device__ __forceinline__ void rng_reopen(ChaChaRng &rng, uint64_t stream0, uint64_t stream1, uint32_t block = 0) {
    chacha_set_stream(rng.state, stream0, stream1, block);
    rng.head = rng.tail;
}

__device__ __forceinline__ uint32_t rng_avail(const ChaChaRng &rng) { return rng.tail - rng.head; }

// Checkpoint: guarantees >= 8 words are available.  Call at wave-convergent points, at least
// once per 8 draws.
template <int UNROLL = 1>
__device__ __forceinline__ void rng_fill(ChaChaRng &rng) {
    if (rng.tail - rng.head < 8) {
        uint32_t x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = rng.state[i];
        chacha_rounds<UNROLL>(x);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t lo = x[2 * i] + rng.state[2 * i], hi = x[2 * i + 1] + rng.state[2 * i + 1];
            rng.ring[((rng.tail + i) & (RNG_RING_WORDS - 1)) * rng.ring_stride] =
                static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
        }
        ++rng.state[12];  // 2^32 blocks = 256 GiB per stream; no stream on this path draws more than a few
        rng.tail += 8;
    }
}

__device__ __forceinline__ uint64_t rng_next_u64(ChaChaRng &rng) {
    const uint64_t v = rng.ring[(rng.head & (RNG_RING_WORDS - 1)) * rng.ring_stride];
    ++rng.head;
    return v;
}

// Persistent lanes without a shared counter: every wave owns one contiguous chunk of elements and hands them out
// in order; the lanes that ask in the same step get consecutive numbers (ballot + prefix count, the running count
// stays in a scalar register).  Must be called by the whole wave.  Which lane computes an element never matters:
// every element's randomness is keyed by its own index.
struct WaveChunk {
    size_t base;     // first element of this wave's chunk
    uint32_t len;    // elements in it
    uint32_t next;   // handed out so far (wave-uniform)
};

__device__ __forceinline__ WaveChunk wave_chunk(size_t total, uint32_t per_lane) {
    const size_t wave = static_cast<size_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    WaveChunk c;
    c.base = wave * 64u * per_lane;
    c.len = c.base < total ? static_cast<uint32_t>(std::min<size_t>(64u * static_cast<size_t>(per_lane), total - c.base)) : 0u;
    c.next = 0;
    return c;
}

__device__ __forceinline__ uint32_t wave_take(WaveChunk &c, bool want) {
    const uint64_t m = __ballot(want);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    const uint32_t e = c.next + before;
    c.next += static_cast<uint32_t>(__popcll(m));
    return e;
}

__device__ __forceinline__ double u64_to_open01(uint64_t w) {
    const double scale = 1.0 / 9007199254740992.0;  // 2^-53
    double u = static_cast<double>(w >> 11) * scale;
    if (u <= 0.0) u = scale;
    else if (u >= 1.0) u = 1.0 - scale;
    return u;
}

// needs 2 words.  Box-Muller with detmath.h's log / cos(2 pi u): fixed IEEE operation sequences,
// so the CPU restatement reproduces every normal bit for bit (sqrt is exactly rounded on both sides)
__device__ __forceinline__ double rng_standard_normal(ChaChaRng &rng) {
    double u1 = u64_to_open01(rng_next_u64(rng));
    double u2 = u64_to_open01(rng_next_u64(rng));
    return sqrt(-2.0 * det_log(u1)) * det_cos2pi(u2);
}

__device__ __forceinline__ uint64_t rng_uniform_mod(ChaChaRng &rng, uint64_t q) {
    const uint64_t max = ~0ull;
    const uint64_t threshold = max - (max % q);
    for (uint32_t step = 0;; ++step) {
        if ((step & 7) == 0) rng_fill(rng);
        uint64_t x = rng_next_u64(rng);
        if (x < threshold) return x % q;
    }
}

// ---- Karney's exact discrete Gaussian (algorithm D of arXiv:1303.6257) as a state machine --
// One keystream word per step.  Sequential form it reproduces, draw for draw:
//   trial: k = G();  if !P(k(k-1)) retry;  s = bit;  j = word % ceil(sigma);  x = ...;
//          if x out of range retry;  (k+1) times B(k,x) must hold, else retry;  return s(i0+j)
//   H (prob e^-1/2): a=U; if !(a<1/2) true; loop { b=U; if !(b<a) false; a=U; if !(a<b) true }
//   G: count consecutive H successes (cap 1024);  P(n): n consecutive H successes
//   B(k,x): y=x,n=0; loop { z=U; if !(z<y) stop; r=U; if !(r<(2k+x)/(2k+2)) stop; y=z; if n>4096 stop; ++n }
//           result = n even
static __device__ int64_t sample_integer_karney(ChaChaRng &rng, double mean, double stddev) {
    if (!(stddev > 0.0) || !isfinite(mean) || !isfinite(stddev)) return static_cast<int64_t>(llround(mean));
    const int64_t ceil_std = static_cast<int64_t>(ceil(stddev));
    if (ceil_std <= 0) return static_cast<int64_t>(llround(mean));

    enum { ST_H0 = 0, ST_H1, ST_H2, ST_SIGN, ST_J, ST_B0, ST_B1 };
    int st = ST_H0;
    bool in_p = false;       // H outcomes feed P (true) or G (false)
    int32_t k = 0;           // G's count
    int32_t p_left = 0;      // H successes P still needs
    int32_t b_left = 0;      // B successes still needed
    int32_t bn = 0;          // B's step parity counter
    double ha = 0.0, hb = 0.0, x = 0.0, x0 = 0.0, y = 0.0, zz = 0.0, bthr = 0.0;
    int64_t s = 1, i0 = 0, j = 0, result = 0;
    int iter = 0;
    bool done = false, fallback = false;

    for (uint32_t step = 0; !done; ++step) {
        if ((step & 7) == 0) rng_fill(rng);
        const uint64_t w = rng_next_u64(rng);
        const double u = u64_to_open01(w);
        int hres = -1;      // outcome of H finished this step
        int bres = -1;      // outcome of B finished this step
        bool restart = false;
        switch (st) {
            case ST_H0:
                ha = u;
                if (!(ha < 0.5)) hres = 1; else st = ST_H1;
                break;
            case ST_H1:
                hb = u;
                if (!(hb < ha)) hres = 0; else st = ST_H2;
                break;
            case ST_H2:
                ha = u;
                if (!(ha < hb)) hres = 1; else st = ST_H1;
                break;
            case ST_SIGN: {
                s = (w & 1ull) ? 1 : -1;
                const double di0 = stddev * static_cast<double>(k) + static_cast<double>(s) * mean;
                i0 = static_cast<int64_t>(ceil(di0));
                x0 = (static_cast<double>(i0) - di0) / stddev;
                st = ST_J;
                break;
            }
            case ST_J:
                j = static_cast<int64_t>(w % static_cast<uint64_t>(ceil_std));
                x = x0 + static_cast<double>(j) / stddev;
                if (!(x < 1.0) || (x == 0.0 && s < 0 && k == 0)) {
                    restart = true;
                } else {
                    b_left = k + 1;
                    bthr = (2.0 * static_cast<double>(k) + x) / static_cast<double>(2 * k + 2);
                    y = x;
                    bn = 0;
                    st = ST_B0;
                }
                break;
            case ST_B0:
                zz = u;
                if (!(zz < y)) bres = (bn % 2) == 0; else st = ST_B1;
                break;
            default:  // ST_B1
                if (!(u < bthr)) {
                    bres = (bn % 2) == 0;
                } else {
                    y = zz;
                    if (bn > 4096) bres = (bn % 2) == 0;
                    else { ++bn; st = ST_B0; }
                }
                break;
        }
        if (hres >= 0) {
            if (!in_p) {  // G: count successes
                bool g_done = hres == 0;
                if (hres == 1) {
                    ++k;
                    if (k > 1024) g_done = true;
                }
                if (g_done) {
                    p_left = k * (k - 1);
                    if (p_left == 0) st = ST_SIGN;
                    else { in_p = true; st = ST_H0; }
                } else {
                    st = ST_H0;
                }
            } else {  // P: needs p_left successes
                if (hres == 1) {
                    if (--p_left == 0) st = ST_SIGN; else st = ST_H0;
                } else {
                    restart = true;
                }
            }
        }
        if (bres >= 0) {
            if (bres == 1) {
                if (--b_left == 0) {
                    result = s * (i0 + j);
                    done = true;
                } else {
                    y = x;
                    bn = 0;
                    st = ST_B0;
                }
            } else {
                restart = true;
            }
        }
        if (restart) {
            if (++iter >= (1 << 16)) {
                fallback = true;
                done = true;
            } else {
                k = 0;
                in_p = false;
                st = ST_H0;
            }
        }
    }
    if (fallback) {
        rng_fill(rng);
        result = static_cast<int64_t>(llround(mean + stddev * rng_standard_normal(rng)));
    }
    return result;
}


// Elements per workgroup chunk for the persistent-lane kernels: one resident round of the whole
// chip when the problem is big enough (chunks are consumed dynamically inside a workgroup, so the
// only imbalance left is the last element of each lane).  MXX_HIP_SAMPLER_PER_LANE=n forces n
// elements per lane (tests use it to exercise stream switching at small sizes).
static inline uint32_t sampler_per_lane(size_t total, const void *kernel, int device, int forced) {
    if (forced >= 1) return static_cast<uint32_t>(forced);
    int blocks_per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, 256, 0) != hipSuccess || blocks_per_cu < 1)
        blocks_per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    const size_t lanes = static_cast<size_t>(blocks_per_cu) * cus * 256u;
    const size_t per = (total + lanes - 1) / lanes;
    return static_cast<uint32_t>(per < 1 ? 1 : (per > 4096 ? 4096 : per));
}

// ---- Persistent-lane form of the same sampler ---------------------------------------------------
// A lane works through a list of coefficients; the wave steps all lanes together:
//   every step      : lanes in the cheap states (H*, B*) consume one keystream word;
//   every 4th step  : the expensive transitions (sign + offset draw with its three divisions
//                     and the 64-bit modulo, the normal fallback, and the caller's
//                     "sample finished" action) run for the lanes parked in them;
//   every 8th step  : keystream refill (all live lanes together).
// A lane only consumes words it holds (rng_avail), so the per-stream draw order - and therefore
// every sample - is that of sample_integer_karney above.  A wave now pays max-over-lanes of the
// SUM of the draw counts of its lanes' coefficients (which concentrates around the mean) instead
// of the per-coefficient maximum, and the expensive code runs a quarter as often.
// Cheap states all have the shape "draw u; continue while u < T":
//   H (Bernoulli e^-1/2): T starts at 1/2 and becomes the last draw; the outcome is the parity
//      of the run length (the H0/H1/H2 states of the sequential form collapse into T + parity);
//   B0: T = y, B1: T = (2k+x)/(2k+2) - the two alternating comparisons of B(k, x).
// Every uniform is m * 2^-53 with the integer m = max(word >> 11, 1), so "u < T" is decided
// exactly on integers: m < ceil(T * 2^53) (scaling by 2^53 is exact).  The cheap path is
// therefore integer-only; doubles appear in the expensive transitions alone.
enum { KS_H = 0, KS_B0, KS_B1, KS_SIGN, KS_FALLBACK, KS_DONE, KS_IDLE };

struct KarneyFsm {
    int32_t st, k, p_left, b_left, bn, iter, par;
    bool in_p;
    uint64_t T, xt, bt, zz;  // thresholds / last draw in units of 2^-53: current, x, (2k+x)/(2k+2), z
    double mean, stddev;
    uint64_t cs, magic;  // ceil(stddev) and floor((2^64-1)/cs)
    int64_t result;
};

#define KARNEY_HALF (1ull << 52)

// t in [0, 1] -> ceil(t * 2^53)
__device__ __forceinline__ uint64_t karney_ticks(double t) { return static_cast<uint64_t>(ceil(t * 9007199254740992.0)); }

struct KarneyDivisor {
    uint64_t cs, magic;
    bool degenerate;  // the sampler returns llround(mean) without drawing
};

__host__ __device__ __forceinline__ KarneyDivisor karney_divisor(double stddev) {
    KarneyDivisor d{0, 0, true};
    if (!(stddev > 0.0) || !(stddev <= 1.7976931348623157e308)) return d;  // also rejects NaN and +inf
    const int64_t c = static_cast<int64_t>(ceil(stddev));
    if (c <= 0) return d;
    d.cs = static_cast<uint64_t>(c);
    d.magic = ~0ull / d.cs;
    d.degenerate = false;
    return d;
}

__device__ __forceinline__ void karney_begin(KarneyFsm &f, double mean, double stddev, const KarneyDivisor &d) {
    f.mean = mean;
    f.stddev = stddev;
    if (d.degenerate || !isfinite(mean)) {
        f.result = static_cast<int64_t>(llround(mean));
        f.st = KS_DONE;
        return;
    }
    f.cs = d.cs;
    f.magic = d.magic;
    f.st = KS_H;
    f.T = KARNEY_HALF;
    f.par = 0;
    f.k = 0;
    f.in_p = false;
    f.iter = 0;
}

// one cheap step
__device__ __forceinline__ void karney_light(KarneyFsm &f, ChaChaRng &rng) {
    if (f.st > KS_B1 || rng_avail(rng) == 0) return;
    uint64_t m = rng_next_u64(rng) >> 11;
    m = m ? m : 1;  // u64_to_open01's clamp away from 0 (m < 2^53 always)
    const bool is_h = f.st == KS_H, is_b0 = f.st == KS_B0, is_b1 = f.st == KS_B1;
    // B's step cap (n > 4096) ends the run as a failed comparison would
    const bool cont = (m < f.T) && !(is_b1 && f.bn > 4096);
    f.zz = is_b0 ? m : f.zz;
    if (cont) {
        f.T = is_h ? m : (is_b0 ? f.bt : f.zz);
        f.par ^= is_h ? 1 : 0;
        f.bn += is_b1 ? 1 : 0;
        f.st = is_h ? KS_H : (is_b0 ? KS_B1 : KS_B0);
        return;
    }
    // the run ended: H reports parity-even, B reports n even
    const bool ok = is_h ? (f.par == 0) : ((f.bn & 1) == 0);
    bool restart = false;
    int32_t next = KS_H;
    if (is_h) {
        if (!f.in_p) {  // G counts consecutive successes, then P needs k(k-1) of them
            const int32_t k2 = f.k + (ok ? 1 : 0);
            const bool g_done = !ok || k2 > 1024;
            const int32_t pl = k2 * (k2 - 1);
            f.k = k2;
            f.p_left = pl;
            f.in_p = g_done && pl != 0;
            next = (g_done && pl == 0) ? KS_SIGN : KS_H;
        } else {
            const int32_t pl = f.p_left - 1;
            f.p_left = pl;
            restart = !ok;
            next = pl == 0 ? KS_SIGN : KS_H;
        }
    } else {
        const int32_t bl = f.b_left - 1;
        f.b_left = bl;
        restart = !ok;
        next = bl == 0 ? KS_DONE : KS_B0;
    }
    if (restart) {
        const int32_t it = f.iter + 1;
        f.iter = it;
        f.k = 0;
        f.in_p = false;
        next = it >= (1 << 16) ? KS_FALLBACK : KS_H;
    }
    f.st = next;
    f.T = next == KS_H ? KARNEY_HALF : f.xt;  // a fresh H run, or B(k, x) starting over with y = x
    f.par = 0;
    f.bn = 0;
}

// the expensive transitions; call at wave-convergent service points
__device__ __forceinline__ void karney_heavy(KarneyFsm &f, ChaChaRng &rng) {
    if (f.st == KS_SIGN && rng_avail(rng) >= 2) {
        const uint64_t w1 = rng_next_u64(rng), w2 = rng_next_u64(rng);
        const int64_t s = (w1 & 1ull) ? 1 : -1;
        const double di0 = f.stddev * static_cast<double>(f.k) + static_cast<double>(s) * f.mean;
        const int64_t i0 = static_cast<int64_t>(ceil(di0));
        const double x0 = (static_cast<double>(i0) - di0) / f.stddev;
        uint64_t j = w2 - __umul64hi(w2, f.magic) * f.cs;  // w2 % cs
        while (j >= f.cs) j -= f.cs;
        const double x = x0 + static_cast<double>(static_cast<int64_t>(j)) / f.stddev;
        if (!(x < 1.0) || (x == 0.0 && s < 0 && f.k == 0)) {
            const int32_t it = f.iter + 1;
            f.iter = it;
            f.k = 0;
            f.in_p = false;
            f.st = it >= (1 << 16) ? KS_FALLBACK : KS_H;
            f.T = KARNEY_HALF;
            f.par = 0;
        } else {
            f.b_left = f.k + 1;
            f.xt = karney_ticks(x);
            f.bt = karney_ticks((2.0 * static_cast<double>(f.k) + x) / static_cast<double>(2 * f.k + 2));
            f.T = f.xt;
            f.bn = 0;
            f.result = s * (i0 + static_cast<int64_t>(j));
            f.st = KS_B0;
        }
    } else if (f.st == KS_FALLBACK && rng_avail(rng) >= 2) {
        f.result = static_cast<int64_t>(llround(f.mean + f.stddev * rng_standard_normal(rng)));
        f.st = KS_DONE;
    }
}
