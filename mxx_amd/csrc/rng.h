// rng.h — counter-based ChaCha20 streams and the integer samplers built on them.
//
// Stream keying follows the reference's device RNG so that a sample is a pure
// function of (seed, stream words, domain tag) and column windows commute
// (cuda/src/ChaCha.cu:104-167; keys in cuda/src/matrix/MatrixSampling.cu:239-289,
// cuda/src/matrix/MatrixTrapdoor.cu:234-241,772-779):
//   subkey  = HChaCha20(key = seed words (LE), nonce = domain_tag || stream2)
//   state   = "expand 32-byte k" | subkey | counter64 = stream0 | nonce64 = stream1
//   output  = successive ChaCha20 blocks (64-bit counter), read as 8 LE u64 words each.
// Discrete Gaussians use Karney's exact rejection sampler with the reference's
// iteration caps (cuda/src/matrix/MatrixSampling.cu:30-147): only IEEE double
// compare / add / mul / div / ceil are involved, so the CPU oracle reproduces the
// stream bit for bit (both sides are built with -ffp-contract=off).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

struct ChaChaRng {
    uint32_t state[16];
    uint32_t block[16];
    uint32_t pos;  // next u64 word inside block, 8 = exhausted
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); \
    c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  \
    c += d; b ^= c; b = rotl32(b, 7);

__device__ __forceinline__ void chacha_rounds(uint32_t (&x)[16]) {
#pragma unroll 1
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

__device__ __forceinline__ void rng_init(ChaChaRng &rng, const GpuRngSeed &seed, uint64_t stream0, uint64_t stream1,
                                         uint64_t stream2, uint64_t domain_tag) {
    uint32_t x[16];
    x[0] = 0x61707865u; x[1] = 0x3320646eu; x[2] = 0x79622d32u; x[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[4 + 2 * i] = static_cast<uint32_t>(seed.words[i]);
        x[5 + 2 * i] = static_cast<uint32_t>(seed.words[i] >> 32);
    }
    x[12] = static_cast<uint32_t>(domain_tag);
    x[13] = static_cast<uint32_t>(domain_tag >> 32);
    x[14] = static_cast<uint32_t>(stream2);
    x[15] = static_cast<uint32_t>(stream2 >> 32);
    chacha_rounds(x);  // HChaCha20: no feed-forward, subkey = words 0..3 and 12..15
    rng.state[0] = 0x61707865u; rng.state[1] = 0x3320646eu; rng.state[2] = 0x79622d32u; rng.state[3] = 0x6b206574u;
    rng.state[4] = x[0]; rng.state[5] = x[1]; rng.state[6] = x[2]; rng.state[7] = x[3];
    rng.state[8] = x[12]; rng.state[9] = x[13]; rng.state[10] = x[14]; rng.state[11] = x[15];
    rng.state[12] = static_cast<uint32_t>(stream0);
    rng.state[13] = static_cast<uint32_t>(stream0 >> 32);
    rng.state[14] = static_cast<uint32_t>(stream1);
    rng.state[15] = static_cast<uint32_t>(stream1 >> 32);
    rng.pos = 8;
}

__device__ __forceinline__ uint64_t rng_next_u64(ChaChaRng &rng) {
    if (rng.pos >= 8) {
        uint32_t x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = rng.state[i];
        chacha_rounds(x);
#pragma unroll
        for (int i = 0; i < 16; ++i) rng.block[i] = x[i] + rng.state[i];
        if (++rng.state[12] == 0) ++rng.state[13];
        rng.pos = 0;
    }
    // dynamic indexing of a register array would spill: select with a static unroll
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (static_cast<uint32_t>(i) == rng.pos) {
            lo = rng.block[2 * i];
            hi = rng.block[2 * i + 1];
        }
    }
    ++rng.pos;
    return static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32);
}

// uniform double in (0,1) from the top 53 bits
__device__ __forceinline__ double rng_uniform_open01(ChaChaRng &rng) {
    const double scale = 1.0 / 9007199254740992.0;  // 2^-53
    double u = static_cast<double>(rng_next_u64(rng) >> 11) * scale;
    if (u <= 0.0) u = scale;
    else if (u >= 1.0) u = 1.0 - scale;
    return u;
}

__device__ __forceinline__ double rng_standard_normal(ChaChaRng &rng) {
    const double two_pi = 6.283185307179586476925286766559;
    double u1 = rng_uniform_open01(rng);
    double u2 = rng_uniform_open01(rng);
    return sqrt(-2.0 * log(u1)) * cos(two_pi * u2);
}

__device__ __forceinline__ uint64_t rng_uniform_mod(ChaChaRng &rng, uint64_t q) {
    const uint64_t max = ~0ull;
    const uint64_t threshold = max - (max % q);
    for (;;) {
        uint64_t x = rng_next_u64(rng);
        if (x < threshold) return x % q;
    }
}

// ---- Karney's exact discrete Gaussian (algorithm D of arXiv:1303.6257) -----------------
// H: true with probability exp(-1/2)
__device__ __forceinline__ bool karney_h(ChaChaRng &rng) {
    double a = rng_uniform_open01(rng);
    if (!(a < 0.5)) return true;
    for (;;) {
        double b = rng_uniform_open01(rng);
        if (!(b < a)) return false;
        a = rng_uniform_open01(rng);
        if (!(a < b)) return true;
    }
}

__device__ __forceinline__ int32_t karney_g(ChaChaRng &rng) {
    int32_t n = 0;
    while (karney_h(rng)) {
        ++n;
        if (n > 1024) break;
    }
    return n;
}

__device__ __forceinline__ bool karney_p(ChaChaRng &rng, int32_t n) {
    while (n-- && karney_h(rng)) {
    }
    return n < 0;
}

__device__ __forceinline__ bool karney_b(ChaChaRng &rng, int32_t k, double x) {
    double y = x;
    int32_t n = 0;
    const double m = static_cast<double>(2 * k + 2);
    for (;; ++n) {
        double z = rng_uniform_open01(rng);
        if (!(z < y)) break;
        double r = rng_uniform_open01(rng);
        if (!(r < (2.0 * static_cast<double>(k) + x) / m)) break;
        y = z;
        if (n > 4096) break;
    }
    return (n % 2) == 0;
}

static __device__ int64_t sample_integer_karney(ChaChaRng &rng, double mean, double stddev) {
    if (!(stddev > 0.0) || !isfinite(mean) || !isfinite(stddev)) return static_cast<int64_t>(llround(mean));
    const int64_t ceil_std = static_cast<int64_t>(ceil(stddev));
    if (ceil_std <= 0) return static_cast<int64_t>(llround(mean));
    for (int iter = 0; iter < (1 << 16); ++iter) {
        int32_t k = karney_g(rng);
        if (!karney_p(rng, k * (k - 1))) continue;
        int64_t s = (rng_next_u64(rng) & 1ull) ? 1 : -1;
        double di0 = stddev * static_cast<double>(k) + static_cast<double>(s) * mean;
        int64_t i0 = static_cast<int64_t>(ceil(di0));
        double x0 = (static_cast<double>(i0) - di0) / stddev;
        int64_t j = static_cast<int64_t>(rng_next_u64(rng) % static_cast<uint64_t>(ceil_std));
        double x = x0 + static_cast<double>(j) / stddev;
        if (!(x < 1.0) || (x == 0.0 && s < 0 && k == 0)) continue;
        int32_t h = k + 1;
        while (h-- > 0 && karney_b(rng, k, x)) {
        }
        if (h >= 0) continue;
        return s * (i0 + j);
    }
    return static_cast<int64_t>(llround(mean + stddev * rng_standard_normal(rng)));
}
