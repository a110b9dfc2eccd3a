// rng.h — counter-based ChaCha20 streams and the integer samplers built on them.
//
// Stream keying follows the reference's device RNG so that a sample is a pure
// function of (seed, stream words, domain tag) and column windows commute
// (cuda/src/ChaCha.cu:104-167; keys in cuda/src/matrix/MatrixSampling.cu:239-289,
// cuda/src/matrix/MatrixTrapdoor.cu:234-241,772-779):
//   subkey  = HChaCha20(key = seed words (LE), nonce = domain_tag || stream2)
//   state   = "expand 32-byte k" | subkey | counter32 = 0 | nonce96 = (stream0, stream1)
//   output  = successive ChaCha20 blocks (32-bit block counter).
// Departures from the reference, deliberate (no reference test pins keystream bytes; the properties callers rely
// on - pure function of (seed, global index), column windows commute, src/sampler/gpu.rs:292-361 - are unchanged):
//  1. ChaCha.cu:138-149 puts stream0 into the block counter words, so block b of stream (s0, s1) is block 0 of
//     stream (s0 + b, s1) - and stream ids are consecutive everywhere (poly+1, column+1, tower+1), i.e. neighbouring
//     polynomials, columns and towers would share keystream one block apart.  Here the counter is a pure block
//     counter and the stream words live in the 96-bit nonce (RFC 8439 layout): stream0 in words 13 and 15[0:16],
//     stream1 in words 14 and 15[16:32] (48 bits each).
//  2. uniform / bit / ternary: eight fixed-position draws per block (sampling.hip).
//  3. (round 3) The Gaussian samplers read the keystream as little-endian 16-BIT DRAWS, 32 per block.  Karney's
//     uniform deviates are only ever compared, so a deviate is drawn lazily: m 2^-53 with m = max(hi 2^37 + lo, 1),
//     hi = one draw, lo (37 bits = three more draws) only when a comparison ties on hi (one comparison in 65536) -
//     first the threshold's lo if the threshold is itself a deviate whose lo is still undrawn, then the deviate's.
//     Every comparison is the one the reference makes on 53-bit deviates; the keystream per integer drops from ~19
//     64-bit words to ~21 draws, i.e. from 2.4 blocks to 0.65.  A 64-bit word (Box-Muller, the offset j of a trial) is
//     four consecutive draws.  The G-sampler's streams are keyed under ONE sub-key per call (coefficient index moved
//     from stream2 into stream0), so no kernel on this path carries a per-lane key.  The Gaussian matrix sampler draws
//     the two coefficients of a pair from one stream (sampling.hip).
// Discrete Gaussians use Karney's exact rejection sampler with the reference's
// iteration caps (cuda/src/matrix/MatrixSampling.cu:30-147): only IEEE double
// compare / add / mul / div / ceil are involved, so the CPU oracle reproduces the
// stream bit for bit (both sides are built with -ffp-contract=off).
//
// Wave64 shape of the sampler.  A rejection sampler written as nested data-dependent
// loops makes a wave pay the product of the per-level maxima over its 64 lanes, and a
// ChaCha block refill inside such loops is executed once per straggling lane.  Here
//   * every lane owns a ring of 64 draws (two blocks) in LDS; blocks are generated only at explicit,
//     wave-convergent checkpoints (rng_fill_wave): when few lanes are short, FOUR LANES COMPUTE ONE BLOCK (a
//     column of the state each, the diagonal rounds through DPP quad permutes), 16 blocks per pass of ~330
//     instructions for whichever lanes asked; when most are, every lane computes its own;
//   * Karney's algorithm is a flat state machine that consumes exactly one draw per step, so all lanes of a wave
//     sit at the same program point and a wave pays max-over-lanes of the TOTAL draw count once.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <cstdlib>

#include "detmath.h"

struct ChaChaRng {
    uint32_t state[16];
    uint32_t *ring;    // this lane's ring: 32-bit slot s (draws 2s, 2s+1) lives at ring[s * ring_stride]
    uint32_t ring_stride;
    uint32_t head;     // draws consumed
    uint32_t tail;     // draws generated (a multiple of 32: blocks land in one half of the ring or the other)
};

// a lane's ring: 32 slots of 32 bits = 64 draws = two keystream blocks (128 bytes, 32 KB per 256 lanes)
#define RNG_RING_SLOTS 32
#ifndef RNG_DRAW_BITS
#define RNG_DRAW_BITS 16  // width of a draw: 16 (shipped; the CPU restatement's) or 8 (timing experiments)
#endif
#define RNG_DRAW_LOG (RNG_DRAW_BITS == 16 ? 1 : 2)          // log2 of the draws per 32-bit ring slot
#define RNG_DRAW_MASK ((1u << RNG_DRAW_BITS) - 1u)
#define RNG_BLOCK_DRAWS (512u / RNG_DRAW_BITS)               // draws per keystream block
#define RNG_U64_DRAWS (64 / RNG_DRAW_BITS)
// shape of a superstep of the persistent-lane kernels: KARNEY_SERVICES x [service point, KARNEY_LIGHTS cheap steps]
// between two keystream checkpoints.  4 x 2 is the measured optimum (same-box sweep of 2x4, 3x2, 3x3, 4x2, 6x2, 8x1,
// 12x1, 16x1 in profiles/r04_notes.md: more cheap steps per service point make the parked lanes wait longer than the
// saved service executions are worth, fewer execute the service blocks - which the whole wave pays for - too often)
#ifndef KARNEY_LIGHTS
#define KARNEY_LIGHTS 4
#endif
#ifndef KARNEY_SERVICES
#define KARNEY_SERVICES 2
#endif
#define KARNEY_SUPERSTEP (KARNEY_LIGHTS * KARNEY_SERVICES)
// Launches with at most one element per lane (the small rings of the GGH15 chain, a few target columns) last as long as
// their unluckiest lane's chain of dependent steps, and almost every service block of such a wave serves one or two lanes:
// they take the superstep with ONE service point per KARNEY_SUPERSTEP cheap steps (same-box: M4 step 0.592 -> 0.567 ms,
// 1-2 target columns of M3A 0.61 / 0.78 -> 0.58 / 0.74 ms; large launches lose 2-3 % with it).  SV = services per
// superstep is a template parameter of the lane kernels; the launchers pick it from per_lane.
constexpr uint32_t karney_urgent(int services) {
    return KARNEY_SUPERSTEP + 6u * services > 31u ? 31u : KARNEY_SUPERSTEP + 6u * services;
}
// a lane is "urgent" when the draws it holds may not last to the next checkpoint: a superstep's cheap steps take one
// draw each, a service point at most six
#define RNG_URGENT (RNG_DRAW_BITS == 16 ? (KARNEY_SUPERSTEP + 6u * KARNEY_SERVICES > 31u ? 31u : KARNEY_SUPERSTEP + 6u * KARNEY_SERVICES) : 30u)
#define RNG_STARVING (RNG_DRAW_BITS == 16 ? 6u : 12u)

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

// stride: lanes per workgroup when the caller knows it at compile time (the persistent-lane kernels pass SAMPLER_THREADS:
// slot addresses then fold into the LDS instructions' immediate offsets), 0 = blockDim.x
__device__ __forceinline__ void rng_init_keyed(ChaChaRng &rng, uint32_t *ring_base, const ChaChaKey &key,
                                               uint64_t stream0, uint64_t stream1, uint32_t stride = 0) {
    rng.state[0] = 0x61707865u; rng.state[1] = 0x3320646eu; rng.state[2] = 0x79622d32u; rng.state[3] = 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) rng.state[4 + i] = key.w[i];
    chacha_set_stream(rng.state, stream0, stream1, 0);
    rng.ring = ring_base + threadIdx.x;
    rng.ring_stride = stride ? stride : blockDim.x;
    rng.head = 0;
    rng.tail = 0;
}

__device__ __forceinline__ void rng_init(ChaChaRng &rng, uint32_t *ring_base, const GpuRngSeed &seed, uint64_t stream0,
                                         uint64_t stream1, uint64_t stream2, uint64_t domain_tag) {
    rng_init_keyed(rng, ring_base, chacha_subkey(seed, stream2, domain_tag), stream0, stream1);
}

// Re-key an open generator to another stream of the same family, positioned at keystream block
// `block`; buffered draws are dropped
__device__ __forceinline__ void rng_reopen(ChaChaRng &rng, uint64_t stream0, uint64_t stream1, uint32_t block = 0) {
    chacha_set_stream(rng.state, stream0, stream1, block);
    rng.head = rng.tail;
}

__device__ __forceinline__ uint32_t rng_avail(const ChaChaRng &rng) { return rng.tail - rng.head; }

// this lane computes the next block of its own stream into the free half of its ring (needs avail <= 32)
template <int UNROLL = 1>
__device__ __forceinline__ void rng_block_lane(ChaChaRng &rng) {
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = rng.state[i];
    chacha_rounds<UNROLL>(x);
    const uint32_t slot0 = (rng.tail >> RNG_DRAW_LOG) & (RNG_RING_SLOTS - 1);  // 0 or 16
#pragma unroll
    for (int i = 0; i < 16; ++i) rng.ring[(slot0 + i) * rng.ring_stride] = x[i] + rng.state[i];
    ++rng.state[12];  // 2^32 blocks = 256 GiB per stream; no stream on this path draws more than a few
    rng.tail += RNG_BLOCK_DRAWS;
}

// Per-lane checkpoint (divergent callers: the one-thread-per-element kernels): afterwards at least 32 draws are
// available.  Call at least once per 8 state-machine steps (a step takes one draw, a service point at most six).
template <int UNROLL = 1>
__device__ __forceinline__ void rng_fill_lane(ChaChaRng &rng) {
    if (rng.tail - rng.head <= RNG_BLOCK_DRAWS) rng_block_lane<UNROLL>(rng);
}

// dpp quad permutes: lane c of every quad reads lane (c + 1) & 3 / (c + 2) & 3 / (c + 3) & 3 of its quad
#define RNG_QUAD_ROT1(v) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x39, 0xf, 0xf, false))
#define RNG_QUAD_ROT2(v) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4e, 0xf, 0xf, false))
#define RNG_QUAD_ROT3(v) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x93, 0xf, 0xf, false))
#define RNG_QUAD_LANE0(v) static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x00, 0xf, 0xf, false))

// Wave-convergent checkpoint of the persistent-lane kernels: EVERY lane of the wave calls it (idle lanes with
// live = false); call once per 8 steps.  A live lane with room for a block (at most 32 draws left) "wants" one; it
// is "urgent" from 20 draws down (8 steps + two service points take at most 19) and "starving" below 6.
// A pass happens when the checkpoint is `scheduled` and some lane is urgent, or when 16 lanes are starving; in a pass
// every wanting lane computes its own block (960 instructions for up to 64 blocks).  A lane that runs dry between
// passes waits (it only consumes draws it holds), so holding passes to every second or third checkpoint trades a few
// idle lane-steps for whole passes: the kernels choose `scheduled` (G-sampler every third checkpoint - an element
// starts with block 0's leftover draws -, p1 every second, the Gaussian matrix - a pair of coefficients per stream -
// every third with starve_limit out of reach: sampling.hip).
// `make FILL_POLICY=1` builds the cooperative form instead (measured, not the default:
// profiles/r03_notes.md): for passes with fewer than 40 wanting lanes the first 16 get a quad
// each; lane c of the quad holds column c of the requester's state (its counter / nonce words fetched with
// ds_bpermute, key and constants are uniform), column rounds in place, diagonal rounds with rows b, c, d rotated by
// 1, 2, 3 lanes through DPP, and writes its four output words into the requester's ring (same bytes either way;
// the key words rng.state[0..11] must be wave-uniform).
#ifndef MXX_FILL_POLICY
#define MXX_FILL_POLICY 0
#endif
__device__ __forceinline__ void rng_fill_wave(ChaChaRng &rng, bool live, bool scheduled = true, int starve_limit = 16,
                                              uint32_t urgent = RNG_URGENT) {
#if MXX_FILL_POLICY == 0
    const uint32_t avail = rng.tail - rng.head;
    const bool pass = (scheduled && __any(live && avail <= urgent)) || __popcll(__ballot(live && avail < RNG_STARVING)) >= starve_limit;
    if (pass && live && avail <= RNG_BLOCK_DRAWS) rng_block_lane<10>(rng);
#else
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t col = lane & 3u;
    for (;;) {
        const uint32_t avail = rng.tail - rng.head;
        const bool want = live && avail <= RNG_BLOCK_DRAWS;
        const uint64_t mw = __ballot(want);
        const uint32_t n = static_cast<uint32_t>(__popcll(mw));
        if (n == 0) break;
        if (n >= 40) {
            if (want) rng_block_lane<10>(rng);
            break;
        }
        if (n < 16 && !__any(live && avail <= urgent)) break;
        // requester with rank r < 16 announces itself to lane 4r (everyone else writes to an odd lane nobody reads)
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mw >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mw), 0u));
        const bool served = want && rank < 16u;
        const uint32_t got = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>((served ? rank * 4u : (lane | 1u)) * 4u), static_cast<int>(lane + 1u)));
        const uint32_t req1 = RNG_QUAD_LANE0(got);  // requester's lane + 1, 0: this quad has no work
        const uint32_t req = req1 ? req1 - 1u : lane;
        const int src = static_cast<int>(req * 4u);
        const uint32_t s12 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(rng.state[12])));
        const uint32_t s13 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(rng.state[13])));
        const uint32_t s14 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(rng.state[14])));
        const uint32_t s15 = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(rng.state[15])));
        const uint32_t rtail = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(src, static_cast<int>(rng.tail)));
        const uint32_t a0 = col == 0 ? rng.state[0] : col == 1 ? rng.state[1] : col == 2 ? rng.state[2] : rng.state[3];
        const uint32_t b0 = col == 0 ? rng.state[4] : col == 1 ? rng.state[5] : col == 2 ? rng.state[6] : rng.state[7];
        const uint32_t c0 = col == 0 ? rng.state[8] : col == 1 ? rng.state[9] : col == 2 ? rng.state[10] : rng.state[11];
        const uint32_t d0 = col == 0 ? s12 : col == 1 ? s13 : col == 2 ? s14 : s15;
        uint32_t a = a0, b = b0, c = c0, d = d0;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            CHACHA_QR(a, b, c, d)
            b = RNG_QUAD_ROT1(b); c = RNG_QUAD_ROT2(c); d = RNG_QUAD_ROT3(d);
            CHACHA_QR(a, b, c, d)
            b = RNG_QUAD_ROT3(b); c = RNG_QUAD_ROT2(c); d = RNG_QUAD_ROT1(d);
        }
        if (req1) {
            uint32_t *dst = rng.ring + (static_cast<int>(req) - static_cast<int>(lane));  // the requester's ring (same wave)
            const uint32_t slot0 = ((rtail >> RNG_DRAW_LOG) & (RNG_RING_SLOTS - 1)) + col;
            dst[slot0 * rng.ring_stride] = a + a0;
            dst[(slot0 + 4u) * rng.ring_stride] = b + b0;
            dst[(slot0 + 8u) * rng.ring_stride] = c + c0;
            dst[(slot0 + 12u) * rng.ring_stride] = d + d0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (served) {
            ++rng.state[12];
            rng.tail += RNG_BLOCK_DRAWS;
        }
    }
#endif
}

// one 16-bit draw (the caller has checked rng_avail)
__device__ __forceinline__ uint32_t rng_next16(ChaChaRng &rng) {
    const uint32_t w = rng.ring[((rng.head >> RNG_DRAW_LOG) & (RNG_RING_SLOTS - 1)) * rng.ring_stride];
    const uint32_t v = (w >> (RNG_DRAW_BITS * (rng.head & ((1u << RNG_DRAW_LOG) - 1u)))) & RNG_DRAW_MASK;
    ++rng.head;
    return v;
}

// four draws, little-endian
__device__ __forceinline__ uint64_t rng_next_u64(ChaChaRng &rng) {
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < RNG_U64_DRAWS; ++i) v |= static_cast<uint64_t>(rng_next16(rng)) << (RNG_DRAW_BITS * i);
    return v;
}

// A 16-bit draw followed by a 64-bit one (the sign / offset pair of a Karney trial; the caller has checked that five draws
// are available): the same five draws as rng_next16 + rng_next_u64, fetched as three ring words and funnel-shifted into
// place - 23 vector instructions instead of 43.
__device__ __forceinline__ void rng_next16_u64(ChaChaRng &rng, uint32_t &w1, uint64_t &w2) {
#if RNG_DRAW_BITS == 16
    const uint32_t s0 = (rng.head >> 1) & (RNG_RING_SLOTS - 1);
    const uint32_t s1 = (s0 + 1u) & (RNG_RING_SLOTS - 1), s2 = (s0 + 2u) & (RNG_RING_SLOTS - 1);
    const uint32_t W0 = rng.ring[s0 * rng.ring_stride], W1 = rng.ring[s1 * rng.ring_stride], W2 = rng.ring[s2 * rng.ring_stride];
    const uint32_t sh = (rng.head & 1u) << 4;  // the run starts in the low or the high half of W0
    const uint32_t a = __builtin_amdgcn_alignbit(W1, W0, sh), b = __builtin_amdgcn_alignbit(W2, W1, sh), c = W2 >> sh;
    w1 = a & 0xffffu;
    w2 = static_cast<uint64_t>(__builtin_amdgcn_alignbit(b, a, 16)) | (static_cast<uint64_t>(__builtin_amdgcn_alignbit(c, b, 16)) << 32);
    rng.head += 1 + RNG_U64_DRAWS;
#else
    w1 = rng_next16(rng);
    w2 = rng_next_u64(rng);
#endif
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
template <typename RNG>
__device__ __forceinline__ double rng_standard_normal(RNG &rng) {
    double u1 = u64_to_open01(rng_next_u64(rng));
    double u2 = u64_to_open01(rng_next_u64(rng));
    return sqrt(-2.0 * det_log(u1)) * det_cos2pi(u2);
}

// ---- Karney's exact discrete Gaussian (algorithm D of arXiv:1303.6257) as a state machine --
// One draw per step.  Sequential form it reproduces, draw for draw (U = a lazy deviate, see the header):
//   trial: k = G();  if !P(k(k-1)) retry;  s = draw & 1;  j = (four draws as a u64) % ceil(sigma);  x = ...;
//          if x out of range retry;  (k+1) times B(k,x) must hold, else retry;  return s(i0+j)
//   H (prob e^-1/2): a=U; if !(a<1/2) true; loop { b=U; if !(b<a) false; a=U; if !(a<b) true }
//   G: count consecutive H successes (cap 1024);  P(n): n consecutive H successes
//   B(k,x): y=x,n=0; loop { z=U; if !(z<y) stop; r=U; if !(r<(2k+x)/(2k+2)) stop; y=z; if n>4096 stop; ++n }
//           result = n even
//
// Persistent-lane form.  A lane works through a list of coefficients; the wave steps all lanes together:
//   every step      : lanes in the cheap states (H, B0, B1) consume one draw;
//   every 4th step  : the expensive transitions (sign + offset draw with its three divisions and the 64-bit
//                     modulo, a tied comparison's low bits, the normal fallback, and the caller's "sample
//                     finished" action) run for the lanes parked in them;
//   every 8th step  : keystream refill (rng_fill_wave).
// A lane only consumes draws it holds (rng_avail), so the per-stream draw order - and therefore every sample - is
// that of the sequential form.  A wave pays max-over-lanes of the SUM of the draw counts of its lanes' coefficients
// (which concentrates around the mean) instead of the per-coefficient maximum.
// Cheap states all have the shape "draw u; continue while u < T":
//   H (Bernoulli e^-1/2): T starts at 1/2 and becomes the last draw; the outcome is the parity of the run length;
//   B0: T = y, B1: T = (2k+x)/(2k+2) - the two alternating comparisons of B(k, x).
// A threshold is ceil(T 2^53) split as (hi = top 16 bits, up to 65536 for T = 1; lo = 37 bits); the cheap path
// compares the draw with hi alone and parks the lane in KS_TIE when they are equal.
// State kept small on purpose (the cheap step is ~75 % of these kernels' instructions outside the generator):
//   st   : KS_H or KS_B for the two kinds of run, else a parked / finished state;
//   cnt  : continues so far in the current run.  H: its parity is the outcome.  B: the two comparisons alternate, so
//          cnt even = "z < y" (B0), cnt odd = "r < (2k+x)/(2k+2)" (B1), n = cnt >> 1;
//   T_hi : top part of the current threshold; zz_hi: top part of B's last z.
// Where a threshold's low bits live is NOT tracked per step; a tie works it out from (st, cnt): the constants 1/2, x,
// (2k+x)/(2k+2) at the start of a run / in B1, otherwise the previous deviate - whose low bits exist only if the
// comparison that drew it tied too, which the tie path records as T_own_cnt / zz_own_cnt (the value of cnt they are
// valid for; -1 at every run start).
enum { KS_H = 0, KS_B, KS_SIGN, KS_TIE, KS_FALLBACK, KS_DONE, KS_IDLE };

struct KarneyFsm {
    int32_t st, k, p_left, b_left, iter;
    uint32_t cnt;
    bool in_p;
    uint32_t T_hi, zz_hi;      // current threshold, B's last z (top parts)
    uint32_t xt_hi, bt_hi;     // x and (2k+x)/(2k+2) in ticks of 2^-53, top parts
    uint32_t tie_h, tie_st;    // the draw that tied and the state it tied in
    uint32_t T_own_cnt, zz_own_cnt;
    uint64_t T_lo, zz_lo;
    double x;                  // the trial's x: the low bits of its two constant thresholds are rebuilt from it when a tie needs them
    double mean, stddev;
    uint64_t cs, magic;  // ceil(stddev) and floor((2^64-1)/cs)
    int64_t result;
};

#define KARNEY_LO_BITS (53 - RNG_DRAW_BITS)
#define KARNEY_LO_MASK ((1ull << KARNEY_LO_BITS) - 1)
#define KARNEY_HALF_HI (1u << (RNG_DRAW_BITS - 1))
#define KARNEY_NO_OWN 0xffffffffu

// t in [0, 1] -> ceil(t * 2^53)
__device__ __forceinline__ uint64_t karney_ticks(double t) { return static_cast<uint64_t>(ceil(t * 9007199254740992.0)); }
// the top part of the same: ticks >> 37 without leaving double precision (both scalings are exact; the conversion truncates)
__device__ __forceinline__ uint32_t karney_ticks_hi(double t) {
    return static_cast<uint32_t>(ldexp(ceil(t * 9007199254740992.0), -KARNEY_LO_BITS));
}
__device__ __forceinline__ double karney_b1_threshold(int32_t k, double x) {
    return (2.0 * static_cast<double>(k) + x) / static_cast<double>(2 * k + 2);
}

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

// once per lane, before the first karney_begin (the select-style transitions read every field)
__device__ __forceinline__ void karney_reset(KarneyFsm &f) {
    f.st = KS_DONE;
    f.k = f.p_left = f.b_left = f.iter = 0;
    f.cnt = 0;
    f.in_p = false;
    f.T_hi = f.zz_hi = f.xt_hi = f.bt_hi = f.tie_h = f.tie_st = 0;
    f.T_own_cnt = f.zz_own_cnt = KARNEY_NO_OWN;
    f.T_lo = f.zz_lo = 0;
    f.x = 0.0;
    f.mean = f.stddev = 0.0;
    f.cs = 1;
    f.magic = 0;
    f.result = 0;
}

// a fresh H run
__device__ __forceinline__ void karney_open_h(KarneyFsm &f) {
    f.st = KS_H;
    f.cnt = 0;
    f.T_hi = KARNEY_HALF_HI;
    f.T_own_cnt = f.zz_own_cnt = KARNEY_NO_OWN;
}

// karney_begin for a sampler whose centre and width never change (the Gaussian matrix): the caller sets mean, stddev, cs
// and magic ONCE after karney_reset (karney_fix_parameters) and they stay wave-uniform loop invariants - scalar registers -
// instead of being re-assigned, lane by lane, at every integer
__device__ __forceinline__ void karney_fix_parameters(KarneyFsm &f, double mean, double stddev, const KarneyDivisor &d) {
    f.mean = mean;
    f.stddev = stddev;
    f.cs = d.cs;
    f.magic = d.magic;
}
__device__ __forceinline__ void karney_begin_fixed(KarneyFsm &f, bool degenerate) {
    if (degenerate) {  // not finite / not positive width or centre: the sequential form returns llround(mean) without drawing
        f.result = static_cast<int64_t>(llround(f.mean));
        f.st = KS_DONE;
        return;
    }
    karney_open_h(f);
    f.k = 0;
    f.in_p = false;
    f.iter = 0;
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
    karney_open_h(f);
    f.k = 0;
    f.in_p = false;
    f.iter = 0;
}

// the transition after a comparison "draw < T" came out as `lt`; h = the draw's top 16 bits, lo = its low 37 bits
// when they were drawn (KNOWN: the comparison tied)
template <bool KNOWN>
__device__ __forceinline__ void karney_advance(KarneyFsm &f, bool lt, uint32_t h, uint64_t lo) {
    const bool is_b = f.st == KS_B, odd = (f.cnt & 1u) != 0;
    const bool is_b0 = is_b && !odd, is_b1 = is_b && odd;
    // B's step cap (n > 4096) ends the run as a failed comparison would
    const bool cont = lt && !(is_b1 && (f.cnt >> 1) > 4096u);
    if (KNOWN) {  // the deviate's low bits exist: remember them for the comparison that may need them
        if (is_b0) { f.zz_lo = lo; f.zz_own_cnt = f.cnt; }
        if (!is_b) { f.T_lo = lo; f.T_own_cnt = f.cnt + 1; }
    }
    const uint32_t zz_prev = f.zz_hi;
    f.zz_hi = is_b0 ? h : zz_prev;
    if (cont) {
        // H: T = this draw;  B0: T = (2k+x)/(2k+2);  B1: T = y = z
        f.T_hi = is_b ? (odd ? zz_prev : f.bt_hi) : h;
        f.cnt += 1;
        return;
    }
    // the run ended: H reports an even number of continues, B reports n = cnt >> 1 even
    const bool ok = ((is_b ? f.cnt >> 1 : f.cnt) & 1u) == 0;
    bool restart = false;
    int32_t next = KS_H;
    if (!is_b) {
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
        next = bl == 0 ? KS_DONE : KS_B;
    }
    if (restart) {
        const int32_t it = f.iter + 1;
        f.iter = it;
        f.k = 0;
        f.in_p = false;
        next = it >= (1 << 16) ? KS_FALLBACK : KS_H;
    }
    f.st = next;
    // a fresh H run, or B(k, x) starting over with y = x
    f.cnt = 0;
    f.T_hi = next == KS_H ? KARNEY_HALF_HI : f.xt_hi;
    f.T_own_cnt = f.zz_own_cnt = KARNEY_NO_OWN;
}

// one cheap step
template <typename RNG>
__device__ __forceinline__ void karney_light(KarneyFsm &f, RNG &rng) {
    if (f.st > KS_B || rng_avail(rng) == 0) return;
    const uint32_t h = rng_next16(rng);
    if (h == f.T_hi) {  // one comparison in 65536: the low bits decide, at the next service point
        f.tie_h = h;
        f.tie_st = static_cast<uint32_t>(f.st);
        f.st = KS_TIE;
        return;
    }
    karney_advance<false>(f, h < f.T_hi, h, 0);
}

// the 37 low bits of a deviate: three draws
template <typename RNG>
__device__ __forceinline__ uint64_t karney_draw_lo(RNG &rng, uint32_t hi) {
    uint64_t bits = 0;  // the smallest whole number of draws that covers the low bits, most significant draw first
    constexpr int kDraws = (KARNEY_LO_BITS + RNG_DRAW_BITS - 1) / RNG_DRAW_BITS;
#pragma unroll
    for (int i = 0; i < kDraws; ++i) bits = (bits << RNG_DRAW_BITS) | rng_next16(rng);
    const uint64_t lo = bits >> (kDraws * RNG_DRAW_BITS - KARNEY_LO_BITS);
    return (hi == 0 && lo == 0) ? 1 : lo;  // a deviate is never 0
}

// the expensive transitions; call at wave-convergent service points
template <typename RNG>
__device__ __forceinline__ void karney_heavy(KarneyFsm &f, RNG &rng) {
    if (f.st == KS_SIGN && rng_avail(rng) >= 1 + RNG_U64_DRAWS) {
        uint32_t w1;
        uint64_t w2;
        rng_next16_u64(rng, w1, w2);
        // s = +-1.  s * mean is +-mean exactly and (double)(int64)ceil(d) is ceil(d) itself (|d| < 2^63): the same values as
        // the sequential form's `s * mean` and `(double)i0` without an int64 <-> double round trip in the service block
        const bool neg = (w1 & 1u) == 0;
        const double di0 = f.stddev * static_cast<double>(f.k) + (neg ? -f.mean : f.mean);
        const double ci0 = ceil(di0);
        const double x0 = (ci0 - di0) / f.stddev;
        uint64_t j = w2 - __umul64hi(w2, f.magic) * f.cs;  // w2 % cs
        while (j >= f.cs) j -= f.cs;
        const double x = x0 + static_cast<double>(static_cast<int64_t>(j)) / f.stddev;
        if (!(x < 1.0) || (x == 0.0 && neg && f.k == 0)) {
            const int32_t it = f.iter + 1;
            f.iter = it;
            f.k = 0;
            f.in_p = false;
            karney_open_h(f);
            if (it >= (1 << 16)) f.st = KS_FALLBACK;
        } else {
            f.b_left = f.k + 1;
            // only the top 16 bits of the two constant thresholds are needed until a comparison ties (one in 65536): the low
            // 37 are rebuilt from x in the tie path instead of being extracted and carried here
            f.x = x;
            f.xt_hi = karney_ticks_hi(x);
            f.bt_hi = karney_ticks_hi(karney_b1_threshold(f.k, x));
            f.T_hi = f.xt_hi;
            f.cnt = 0;
            f.T_own_cnt = f.zz_own_cnt = KARNEY_NO_OWN;
            const int64_t mag = static_cast<int64_t>(ci0) + static_cast<int64_t>(j);
            f.result = neg ? -mag : mag;
            f.st = KS_B;
        }
    } else if (f.st == KS_TIE && rng_avail(rng) >= 2 * ((KARNEY_LO_BITS + RNG_DRAW_BITS - 1) / RNG_DRAW_BITS)) {
        // the threshold's low bits first (drawn now if it is a deviate that never needed them), then the deviate's
        const bool is_b = f.tie_st == KS_B, odd = (f.cnt & 1u) != 0;
        uint64_t tlo;
        if (!is_b) {  // H: 1/2 at the start of the run, then the previous deviate
            if (f.cnt == 0) tlo = 0;
            else if (f.T_own_cnt == f.cnt) tlo = f.T_lo;
            else tlo = karney_draw_lo(rng, f.T_hi);
        } else if (odd) {  // B1: (2k+x)/(2k+2)
            tlo = karney_ticks(karney_b1_threshold(f.k, f.x)) & KARNEY_LO_MASK;
        } else {  // B0: x at the start of the run, then y = the z of two comparisons ago
            if (f.cnt == 0) tlo = karney_ticks(f.x) & KARNEY_LO_MASK;
            else if (f.zz_own_cnt == f.cnt - 2) tlo = f.zz_lo;
            else tlo = karney_draw_lo(rng, f.T_hi);
        }
        const uint64_t ulo = karney_draw_lo(rng, f.tie_h);
        f.st = static_cast<int32_t>(f.tie_st);
        karney_advance<true>(f, ulo < tlo, f.tie_h, ulo);
    } else if (f.st == KS_FALLBACK && rng_avail(rng) >= 2 * RNG_U64_DRAWS) {
        f.result = static_cast<int64_t>(llround(f.mean + f.stddev * rng_standard_normal(rng)));
        f.st = KS_DONE;
    }
}

// the same machine run by one lane on its own (the one-thread-per-element kernels)
static __device__ int64_t sample_integer_karney(ChaChaRng &rng, double mean, double stddev) {
    KarneyFsm f;
    karney_reset(f);
    karney_begin(f, mean, stddev, karney_divisor(stddev));
    for (uint32_t step = 0; f.st != KS_DONE; ++step) {
        if ((step & 7) == 0) rng_fill_lane(rng);
        karney_heavy(f, rng);
        karney_light(f, rng);
    }
    return f.result;
}

// Launch shape of the persistent-lane kernels (round 4).  A workgroup is ONE wave (SAMPLER_THREADS = 64) that owns a
// chunk of 64 x per_lane elements, consumed dynamically by its lanes.  per_lane = enough to give every resident lane an
// element, capped at kSamplerPerLaneCap: large problems then run as SEVERAL rounds of small workgroups instead of one
// resident round of long ones.  Same-box sweeps (profiles/r04_notes.md), M3A call: one round of 256-thread workgroups at
// 25 elements per lane 6.84-6.90 ms; capped at 8: 6.60; 64-thread workgroups capped at 8: 6.47-6.50 (cap 6: 6.48-6.50,
// 12: 6.50-6.52, 16: 6.79-6.83); M4 step 0.621 -> 0.592; the 7-column shard (one rank's share of M3A at N = 8) 1.457 ->
// 1.377.  A wave's SIMD slot and LDS go back to the dispatcher as soon as ITS lanes are done, not when the slowest of four
// waves is; and the waves of a round start staggered, which spreads their keystream passes and service blocks in time.
// A persistent one-round grid fed from a global chunk counter (no tail between a wave's chunks) was built and measured:
// slower at the occupancy the runtime reports (6.93 ms: workgroups beyond the really resident ones hold their static first
// chunks until the end), equal when oversubscribed (6.44) - removed.
// MXX_HIP_SAMPLER_PER_LANE=n forces n elements per lane (tests exercise stream switching with it).
#ifndef SAMPLER_PER_LANE_CAP
#define SAMPLER_PER_LANE_CAP 8
#endif
constexpr uint32_t kSamplerPerLaneCap = SAMPLER_PER_LANE_CAP;
// threads per workgroup of the persistent-lane kernels (a lane's ring is 128 bytes of LDS: 8 KB per wave)
#ifndef SAMPLER_THREADS
#define SAMPLER_THREADS 64
#endif
// ---- column segments: several independent requests sampled by ONE launch -------------------------------------------
// The GGH15 callers hand `preimage_batched_sharded` dozens of small requests against one trapdoor
// (src/sampler/trapdoor/gpu.rs:371-397, src/lookup/ggh15/pubkey_gpu.rs:615-971); at n = 256 a request of four columns
// leaves the chip ~97 % idle and lasts as long as its unluckiest lane's chain of dependent Karney steps.  The *_segments
// entry points sample the column-wise concatenation of such requests in one launch: segment j = columns
// [start[j], start[j + 1]) carries its own seed, and every element's stream is keyed by its position INSIDE its segment
// (local column, the segment's own column count), so the columns of segment j equal - bit for bit - what the plain entry
// point writes for a matrix of that segment's shape under that seed.  The table travels as a kernel argument (kernarg
// memory: wave-uniform scalar loads); the lane kernels take it in their UNI form only - a wave's chunk inside one
// polynomial / column - so the key stays in scalar registers exactly as in the single-seed kernels.
#define RNG_MAX_SEGMENTS 64
struct RngSegments {
    uint32_t count;
    uint32_t start[RNG_MAX_SEGMENTS + 1];  // first column of segment j; start[count] = all columns
    ChaChaKey key[RNG_MAX_SEGMENTS];       // chacha_subkey(seed_j, 0, the sampler's domain tag)
};
struct NoSegments {};
template <bool SEG>
using SegArg = typename std::conditional<SEG, RngSegments, NoSegments>::type;

#if defined(__HIPCC__)
// segment of column `col` (binary search; with a wave-uniform `col` everything here runs on the scalar unit)
__device__ __forceinline__ uint32_t rng_segment_of(const RngSegments &s, uint32_t col) {
    uint32_t lo = 0, hi = s.count;  // start[lo] <= col < start[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (col >= s.start[mid]) lo = mid;
        else hi = mid;
    }
    return lo;
}
#endif

// host side: the table of `nseg` segments (seeds, column counts) under a sampler's domain tag; false if it does not fit
static inline bool rng_segments_build(RngSegments &out, const GpuRngSeed *seeds, const size_t *seg_cols, size_t nseg,
                                      uint64_t domain_tag, size_t total_cols) {
    if (nseg == 0 || nseg > RNG_MAX_SEGMENTS || !seeds || !seg_cols) return false;
    size_t at = 0;
    out.count = static_cast<uint32_t>(nseg);
    for (size_t j = 0; j < nseg; ++j) {
        if (seg_cols[j] == 0 || at + seg_cols[j] > 0xffffffffull) return false;
        out.start[j] = static_cast<uint32_t>(at);
        out.key[j] = chacha_subkey(seeds[j], 0, domain_tag);
        at += seg_cols[j];
    }
    for (size_t j = nseg; j <= RNG_MAX_SEGMENTS; ++j) out.start[j] = static_cast<uint32_t>(at);
    for (size_t j = nseg; j < RNG_MAX_SEGMENTS; ++j) out.key[j] = ChaChaKey{};
    return at == total_cols;
}

static inline uint32_t sampler_per_lane(size_t total, const void *kernel, int device, int forced) {
    if (forced >= 1) return static_cast<uint32_t>(forced);
    int blocks_per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, SAMPLER_THREADS, 0) != hipSuccess || blocks_per_cu < 1)
        blocks_per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    const size_t lanes = static_cast<size_t>(blocks_per_cu) * cus * SAMPLER_THREADS;
    const size_t per = (total + lanes - 1) / lanes;
    return static_cast<uint32_t>(per < 1 ? 1 : (per > kSamplerPerLaneCap ? kSamplerPerLaneCap : per));
}

