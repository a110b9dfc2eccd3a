// serde.hip — compact wire format: coefficient-domain, CRT-reconstructed, centred integers,
// bit-packed at the matrix-wide maximum width.  Replaces the compact-bytes half of
// cuda/src/matrix/MatrixSerde.cu (ABI: cuda/include/matrix/MatrixSerde.cuh:35-58).
//
// Format (MatrixSerde.cu:280-456,1535-1627; Rust header src/matrix/gpu_dcrt_poly.rs:956-1044):
//   coefficient idx = poly*N + i occupies bits [idx*w, (idx+1)*w) of a little-endian bit
//   stream; its low w-1 bits are |x| and bit w-1 is the sign, where x is the representative
//   of the coefficient in (-Q/2, Q/2] (negative iff value > floor(Q/2)), Q = q_0..q_level;
//   w = max_coeff_bits = 1 + max bit-width of |x| over the matrix (0 for the zero matrix),
//   bytes_per_coeff = ceil(w/8), payload_len = ceil(polys*N*w/8).
//
// Device pipeline: one thread per coefficient does Garner mixed-radix CRT from the limb
// residues (inverse table built at context creation, as Runtime.cu:77-96), Horner-evaluates
// the multi-word integer, centres it, and (pass 1) contributes to the max bit-width or
// (pass 2) ORs its w bits into the zeroed payload.  Nothing is staged per coefficient in HBM:
// the value is recomputed in pass 2 (O(L^2) modmuls, cheaper than a round trip of L words).
#include "common.h"
#include "modarith.h"

#include <algorithm>

static constexpr int kMaxWords = 64;  // up to 4096-bit Q

struct SerdeConsts {
    int limbs;
    int words;                    // 64-bit words per reconstructed coefficient
    uint64_t q[GPUPOLY_MAX_LIMBS];
    uint64_t modulus[kMaxWords];  // Q, little-endian words
    uint64_t half[kMaxWords];     // floor(Q/2)
};

// residues of coefficient (poly, i) -> |x| words (little-endian), sign; returns bit width of |x|.
// ML bounds the limb count at compile time (8, 16 or 64) so that the mixed-radix digits and the words of x stay in
// registers for the sizes that matter (the unbounded form kept two 64-entry arrays in scratch memory).
// Fast path: the matrices that are serialised most - preimages, trapdoors, Gaussian-sized keys - hold SMALL integers:
// if every limb's residue is the residue of the centred limb-0 value c (|c| <= q_0 / 2), then x = c, by uniqueness of
// the CRT representative in (-Q/2, Q/2]; that is an O(L) comparison instead of the O(L^2) Garner recurrence.
// General path: Garner's products go through Barrett (mu of every limb from the context) instead of a 128-bit `%`.
template <typename W, int ML>
__device__ __forceinline__ uint32_t reconstruct_centered(const W *__restrict__ src, size_t poly, uint32_t i, uint32_t N,
                                                         const SerdeConsts &sc, const uint64_t *__restrict__ garner,
                                                         size_t garner_stride, const LimbConst *__restrict__ limbs,
                                                         uint64_t *x, bool &negative) {
    const int L = sc.limbs, WC = sc.words;
    uint64_t res[ML];
    if constexpr (ML <= 16) {
#pragma unroll
        for (int k = 0; k < ML; ++k)
            if (k < L) res[k] = static_cast<uint64_t>(src[(poly * L + k) * N + i]);
    } else {
        for (int k = 0; k < L; ++k) res[k] = static_cast<uint64_t>(src[(poly * L + k) * N + i]);
    }
    {
        const uint64_t q0 = sc.q[0];
        const bool neg0 = res[0] > (q0 >> 1);
        const uint64_t mag = neg0 ? q0 - res[0] : res[0];  // |c|
        bool small = true;
        auto same = [&](int k) {
            const uint64_t qk = sc.q[k];
            return mag < qk && res[k] == (neg0 ? qk - mag : mag);
        };
        if constexpr (ML <= 16) {
#pragma unroll
            for (int k = 1; k < ML; ++k)
                if (k < L) small = small && same(k);
        } else {
            for (int k = 1; k < L; ++k) small = small && same(k);
        }
        if (small) {
            for (int w = 0; w < WC; ++w) x[w] = 0;
            x[0] = mag;
            negative = neg0 && mag != 0;
            return mag ? 64u - static_cast<uint32_t>(__clzll(mag)) : 0u;
        }
    }
    // Second fast path: |x| < q_0 q_1 / 2 - a preimage's entries (perturbations of width ~2^27 against 24-bit limbs) overflow
    // limb 0 alone but not two limbs.  The candidate is the centred two-limb CRT value c2 (one Garner step, one double-width
    // word); if every further limb holds c2's residue then x = c2, again by uniqueness of the representative in
    // (-Q/2, Q/2] (|c2| <= q_0 q_1 / 2 < Q / 2 from three limbs on; with two limbs c2 IS the general result).  O(L) instead
    // of the O(L^2) recurrence below: the width and pack passes over an M3A preimage went from 3.4 + 2.0 ms to the
    // time of reading the matrix (bench.py, compact_bytes).
    if (L >= 2) {
        typedef typename Wide<W>::type D;
        const uint64_t q0 = sc.q[0], q1 = sc.q[1];
        const uint64_t r0m = res[0] >= q1 ? res[0] % q1 : res[0];
        const uint64_t dd = res[1] >= r0m ? res[1] - r0m : res[1] + q1 - r0m;
        const uint64_t v1 = static_cast<uint64_t>(barrett_reduce(static_cast<D>(dd) * static_cast<D>(garner[garner_stride]), static_cast<W>(q1), limbs[1].mu, limbs[1].kbits));
        const D q01 = static_cast<D>(q0) * q1;
        const D val = static_cast<D>(res[0]) + static_cast<D>(v1) * q0;  // in [0, q_0 q_1)
        const bool neg2 = val > (q01 >> 1);
        const D mag2 = neg2 ? q01 - val : val;
        bool ok = true;
        auto same2 = [&](int k) {
            const uint64_t qk = sc.q[k];
            const uint32_t kb = limbs[k].kbits;
            const bool fits = 2 * kb >= 8 * sizeof(D) || (mag2 >> (2 * kb)) == 0;  // Barrett's range: mag2 < 2^(2 bits(q_k))
            const uint64_t r = static_cast<uint64_t>(barrett_reduce(mag2, static_cast<W>(qk), limbs[k].mu, kb));
            return fits && res[k] == ((neg2 && r) ? qk - r : r);
        };
        if constexpr (ML <= 16) {
#pragma unroll
            for (int k = 2; k < ML; ++k)
                if (k < L) ok = ok && same2(k);
        } else {
            for (int k = 2; k < L; ++k) ok = ok && same2(k);
        }
        if (ok) {
            for (int w = 0; w < WC; ++w) x[w] = 0;
            x[0] = static_cast<uint64_t>(mag2);
            uint64_t hi = 0;
            if constexpr (sizeof(D) > 8) hi = static_cast<uint64_t>(mag2 >> 64);
            if (WC > 1) x[1] = hi;
            negative = neg2 && mag2 != 0;
            if (hi) return 128u - static_cast<uint32_t>(__clzll(hi));
            return x[0] ? 64u - static_cast<uint32_t>(__clzll(x[0])) : 0u;
        }
    }
    uint64_t v[ML];
    // Garner: v_k = (r_k - (v_0 + v_1 q_0 + ...)) / (q_0 .. q_{k-1})  mod q_k, computed incrementally
    auto garner_step = [&](int k, int j, uint64_t t, uint64_t qk, uint64_t mu, uint32_t kb) {
        uint64_t vj = v[j];
        if (vj >= qk) vj %= qk;  // only when an earlier modulus is wider than this one
        const uint64_t d = t >= vj ? t - vj : t + qk - vj;
        const uint64_t g = garner[k * garner_stride + j];
        if constexpr (sizeof(W) == 4) return static_cast<uint64_t>(barrett_reduce(d * g, static_cast<uint32_t>(qk), mu, kb));
        else return barrett_reduce(static_cast<u128_t>(d) * g, qk, mu, kb);
    };
    if constexpr (ML <= 16) {  // fully unrolled, guarded: everything stays in registers
#pragma unroll
        for (int k = 0; k < ML; ++k)
            if (k < L) {
                const uint64_t qk = sc.q[k], mu = limbs[k].mu;
                const uint32_t kb = limbs[k].kbits;
                uint64_t t = res[k];
#pragma unroll
                for (int j = 0; j < ML; ++j)
                    if (j < k) t = garner_step(k, j, t, qk, mu, kb);
                v[k] = t;
            }
    } else {
        for (int k = 0; k < L; ++k) {
            const uint64_t qk = sc.q[k], mu = limbs[k].mu;
            const uint32_t kb = limbs[k].kbits;
            uint64_t t = res[k];
            for (int j = 0; j < k; ++j) t = garner_step(k, j, t, qk, mu, kb);
            v[k] = t;
        }
    }
    // Horner: x = (..(v_{L-1} q_{L-2} + v_{L-2}) q_{L-3} + ..) q_0 + v_0
    for (int w = 0; w < WC; ++w) x[w] = 0;
    auto horner_step = [&](int k) {
        const uint64_t m = sc.q[k];
        u128_t carry = v[k];
        for (int w = 0; w < WC; ++w) {
            const u128_t p = static_cast<u128_t>(x[w]) * m + carry;
            x[w] = static_cast<uint64_t>(p);
            carry = p >> 64;
        }
    };
    if constexpr (ML <= 16) {
#pragma unroll
        for (int k = ML - 1; k >= 0; --k) {
            if (k == L - 1) x[0] = v[k];
            else if (k < L - 1) horner_step(k);
        }
    } else {
        x[0] = v[L - 1];
        for (int k = L - 2; k >= 0; --k) horner_step(k);
    }
    // centre: negative iff x > floor(Q/2)
    int cmp = 0;
    for (int w = WC - 1; w >= 0; --w) {
        if (x[w] != sc.half[w]) {
            cmp = x[w] > sc.half[w] ? 1 : -1;
            break;
        }
    }
    negative = cmp > 0;
    if (negative) {
        uint64_t borrow = 0;
        for (int w = 0; w < WC; ++w) {
            const uint64_t a = sc.modulus[w], b = x[w];
            const uint64_t d = a - b - borrow;
            borrow = (a < b + borrow) || (b + borrow < b) ? 1 : 0;
            x[w] = d;
        }
    }
    for (int w = WC - 1; w >= 0; --w)
        if (x[w]) return static_cast<uint32_t>(w) * 64u + (64u - static_cast<uint32_t>(__clzll(x[w])));
    return 0;
}

// ---- fast-path-only forms ----------------------------------------------------------------------------------------------
// The matrices that get serialised (preimages, trapdoors, Gaussian-sized keys) never leave the two fast paths of
// reconstruct_centered, but kernels that also carry the general Garner path pay for it in registers (102 / 73 VGPRs, the
// word array indexed dynamically in scratch): 0.45 + 0.66 ms for an M3A preimage against 0.09 ms of reading it.  These
// forms hold ONLY the fast paths - a coefficient is |x| < q_0 / 2 or |x| < q_0 q_1 / 2, checked against every further
// limb - and raise a flag for anything else; the host then reruns the general kernels (never, for the matrices above).
template <typename W, int ML>
__device__ __forceinline__ bool reconstruct_small(const W *__restrict__ src, size_t poly, uint32_t i, uint32_t N,
                                                  const SerdeConsts &sc, const uint64_t *__restrict__ garner,
                                                  size_t garner_stride, const LimbConst *__restrict__ limbs,
                                                  uint64_t &mag_lo, uint64_t &mag_hi, bool &negative) {
    static_assert(ML <= 16, "fast forms are unrolled over the limbs");
    typedef typename Wide<W>::type D;
    const int L = sc.limbs;
    uint64_t res[ML];
#pragma unroll
    for (int k = 0; k < ML; ++k)
        if (k < L) res[k] = static_cast<uint64_t>(src[(poly * L + k) * N + i]);
    const uint64_t q0 = sc.q[0];
    {
        const bool neg0 = res[0] > (q0 >> 1);
        const uint64_t mag = neg0 ? q0 - res[0] : res[0];
        bool small = true;
#pragma unroll
        for (int k = 1; k < ML; ++k)
            if (k < L) {
                const uint64_t qk = sc.q[k];
                small = small && mag < qk && res[k] == (neg0 ? qk - mag : mag);
            }
        if (small) {
            mag_lo = mag;
            mag_hi = 0;
            negative = neg0 && mag != 0;
            return true;
        }
    }
    if (L < 2) return false;
    const uint64_t q1 = sc.q[1];
    const uint64_t r0m = res[0] >= q1 ? res[0] % q1 : res[0];
    const uint64_t dd = res[1] >= r0m ? res[1] - r0m : res[1] + q1 - r0m;
    const uint64_t v1 = static_cast<uint64_t>(barrett_reduce(static_cast<D>(dd) * static_cast<D>(garner[garner_stride]), static_cast<W>(q1), limbs[1].mu, limbs[1].kbits));
    const D q01 = static_cast<D>(q0) * q1;
    const D val = static_cast<D>(res[0]) + static_cast<D>(v1) * q0;
    const bool neg2 = val > (q01 >> 1);
    const D mag2 = neg2 ? q01 - val : val;
    bool ok = true;
#pragma unroll
    for (int k = 2; k < ML; ++k)
        if (k < L) {
            const uint64_t qk = sc.q[k];
            const uint32_t kb = limbs[k].kbits;
            const bool fits = 2 * kb >= 8 * sizeof(D) || (mag2 >> (2 * kb)) == 0;
            const uint64_t r = static_cast<uint64_t>(barrett_reduce(mag2, static_cast<W>(qk), limbs[k].mu, kb));
            ok = ok && fits && res[k] == ((neg2 && r) ? qk - r : r);
        }
    mag_lo = static_cast<uint64_t>(mag2);
    mag_hi = 0;
    if constexpr (sizeof(D) > 8) mag_hi = static_cast<uint64_t>(mag2 >> 64);
    negative = neg2 && mag2 != 0;
    return ok;
}

template <typename W, int ML>
__global__ void __launch_bounds__(256) compact_maxbits_fast_kernel(const W *__restrict__ src, size_t polys, uint32_t N, SerdeConsts sc,
                                            const uint64_t *__restrict__ garner, size_t garner_stride,
                                            const LimbConst *__restrict__ limbs, unsigned int *__restrict__ max_and_flag) {
    // grid-stride: a few thousand workgroups keep a running maximum in registers and touch the shared word once each (a
    // read of that ONE word per wave - 280 000 of them for an M3A preimage - was 0.2 of this kernel's 0.32 ms)
    const size_t total = polys * N, stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    unsigned int bits = 0;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        uint64_t lo, hi;
        bool neg;
        if (reconstruct_small<W, ML>(src, idx / N, static_cast<uint32_t>(idx % N), N, sc, garner, garner_stride, limbs, lo, hi, neg))
            bits = max(bits, hi ? 128u - static_cast<uint32_t>(__clzll(hi)) : (lo ? 64u - static_cast<uint32_t>(__clzll(lo)) : 0u));
        else
            max_and_flag[1] = 1u;  // some coefficient needs the general path: the host reruns with the general kernels
    }
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, __shfl_down(bits, off));
    if ((threadIdx.x & 63) == 0 && bits > __atomic_load_n(max_and_flag, __ATOMIC_RELAXED)) atomicMax(max_and_flag, bits);
}

// width <= 130 here (|x| below 2^102 at most); the block's 8 * width payload words are assembled in LDS as in the general form
template <typename W, int ML>
__global__ void __launch_bounds__(256) compact_pack_fast_kernel(const W *__restrict__ src, size_t polys, uint32_t N, SerdeConsts sc,
                                         const uint64_t *__restrict__ garner, size_t garner_stride,
                                         const LimbConst *__restrict__ limbs, uint32_t width, uint32_t *__restrict__ payload_words,
                                         size_t payload_word_count) {
    extern __shared__ uint32_t pack_words[];
    const size_t idx = item_index();
    const size_t block_first = idx - threadIdx.x;
    const uint32_t nwords = 8u * width;
    for (uint32_t w = threadIdx.x; w < nwords; w += 256) pack_words[w] = 0;
    __syncthreads();
    if (idx < polys * N) {
        uint64_t lo, hi;
        bool neg;
        (void)reconstruct_small<W, ML>(src, idx / N, static_cast<uint32_t>(idx % N), N, sc, garner, garner_stride, limbs, lo, hi, neg);
        // 192 bits: |x| in words 0..1, the sign bit at width - 1 (at most bit 129)
        uint64_t x0 = lo, x1 = hi, x2 = 0;
        if (neg) {
            const uint32_t sb = width - 1;
            const uint64_t bit = 1ull << (sb & 63);
            if (sb < 64) x0 |= bit;
            else if (sb < 128) x1 |= bit;
            else x2 |= bit;
        }
        const uint32_t base = threadIdx.x * width;
        uint32_t done = 0;
        while (done < width) {
            const uint32_t bit = base + done;
            const uint32_t off = bit & 31u;
            const uint32_t take = min(32u - off, width - done);
            const uint32_t wi = done >> 6, bo = done & 63u;
            const uint64_t cur = wi == 0 ? x0 : (wi == 1 ? x1 : x2), nxt = wi == 0 ? x1 : (wi == 1 ? x2 : 0ull);
            uint64_t chunk = cur >> bo;
            if (bo + take > 64) chunk |= nxt << (64 - bo);
            const uint32_t val = static_cast<uint32_t>(chunk & ((take == 32) ? 0xffffffffull : ((1ull << take) - 1)));
            if (val) atomicOr(&pack_words[bit >> 5], val << off);
            done += take;
        }
    }
    __syncthreads();
    const size_t first_word = block_first / 256 * nwords;
    for (uint32_t w = threadIdx.x; w < nwords; w += 256)
        if (first_word + w < payload_word_count) payload_words[first_word + w] = pack_words[w];
}

template <typename W, int ML>
__global__ void compact_maxbits_kernel(const W *__restrict__ src, size_t polys, uint32_t N, SerdeConsts sc,
                                       const uint64_t *__restrict__ garner, size_t garner_stride,
                                       const LimbConst *__restrict__ limbs, unsigned int *__restrict__ max_bits) {
    const size_t idx = item_index();
    unsigned int bits = 0;
    if (idx < polys * N) {
        uint64_t x[ML];
        bool neg;
        bits = reconstruct_centered<W, ML>(src, idx / N, static_cast<uint32_t>(idx % N), N, sc, garner, garner_stride, limbs, x, neg);
    }
    // wave-level max; the atomic only when this wave would raise the running maximum.  (One unconditional atomic per wave
    // was 280 000 read-modify-writes of ONE word for an M3A preimage - serialised in L2, 3.2 ms, the whole kernel; after the
    // first waves the plain read sees the final width and nearly every wave skips it.)
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, __shfl_down(bits, off));
    if ((threadIdx.x & 63) == 0 && bits > __atomic_load_n(max_bits, __ATOMIC_RELAXED)) atomicMax(max_bits, bits);
}

// A block of 256 coefficients fills exactly 8 * width payload words (256 * width bits, word-aligned for every width), so
// the block assembles them in LDS - 32-bit ORs on shared memory - and writes them out whole, coalesced: no global atomics
// and no memset of the payload (round 2's form ORed every coefficient's two or three pieces into global words: 0.73 ms
// for an M3A preimage against the 0.09 ms of reading it).  Widths beyond kPackLdsWidth bits per coefficient (Q above
// 2^1023) keep the global-atomic form on a zeroed payload.
static constexpr uint32_t kPackLdsWidth = 1024;  // 8 * 1024 words = 32 KB of LDS
template <typename W, int ML, bool LDS>
__global__ void __launch_bounds__(256) compact_pack_kernel(const W *__restrict__ src, size_t polys, uint32_t N, SerdeConsts sc,
                                    const uint64_t *__restrict__ garner, size_t garner_stride,
                                    const LimbConst *__restrict__ limbs, uint32_t width, uint32_t *__restrict__ payload_words,
                                    size_t payload_word_count) {
    extern __shared__ uint32_t pack_words[];  // LDS form: 8 * width words
    const size_t idx = item_index();
    const size_t block_first = idx - threadIdx.x;
    const uint32_t nwords = 8u * width;
    if constexpr (LDS) {
        for (uint32_t w = threadIdx.x; w < nwords; w += 256) pack_words[w] = 0;
        __syncthreads();
    }
    if (idx < polys * N) {
        uint64_t x[ML + 1];  // words of |x| (at most one per limb) + one for the sign bit / the shifted read below
        bool neg;
        reconstruct_centered<W, ML>(src, idx / N, static_cast<uint32_t>(idx % N), N, sc, garner, garner_stride, limbs, x, neg);
        // set the sign bit at position width-1
        const uint32_t sb = width - 1;
        for (int w = sc.words; w <= ML; ++w) x[w] = 0;
        if (neg) x[sb >> 6] |= 1ull << (sb & 63);
        // OR the `width` bits into the stream at bit offset idx*width, 32 bits at a time
        const size_t base = LDS ? static_cast<size_t>(threadIdx.x) * width : idx * static_cast<size_t>(width);
        uint32_t done = 0;
        while (done < width) {
            const size_t bit = base + done;
            const uint32_t off = static_cast<uint32_t>(bit & 31);
            const uint32_t take = min(32u - off, width - done);
            const uint32_t wi = done >> 6, bo = done & 63;
            uint64_t chunk = x[wi] >> bo;
            if (bo + take > 64) chunk |= x[wi + 1] << (64 - bo);
            const uint32_t val = static_cast<uint32_t>(chunk & ((take == 32) ? 0xffffffffull : ((1ull << take) - 1)));
            if (val) {
                if constexpr (LDS) atomicOr(&pack_words[bit >> 5], val << off);
                else atomicOr(&payload_words[bit >> 5], val << off);
            }
            done += take;
        }
    }
    if constexpr (LDS) {
        __syncthreads();
        const size_t first_word = block_first / 256 * nwords;  // block_first is a multiple of 256
        for (uint32_t w = threadIdx.x; w < nwords; w += 256)
            if (first_word + w < payload_word_count) payload_words[first_word + w] = pack_words[w];
    }
}

// 32 bits of the payload starting at bit `bit` (little-endian bit stream); the payload is padded by 8 bytes
__device__ __forceinline__ uint32_t payload_bits32(const uint8_t *__restrict__ payload, size_t bit) {
    const size_t byte = bit >> 3;
    uint64_t v = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) v |= static_cast<uint64_t>(payload[byte + b]) << (8 * b);
    return static_cast<uint32_t>(v >> (bit & 7));
}

// |x| arrives as `mag_bits` bits; every limb reduces it 32 bits at a time, most significant first:
// r <- (r 2^32 + word) mod q, one multiply-high by floor(2^64 / q) per step for word-sized moduli (the first form
// reduced a 128-bit value with `%` for every byte of every limb)
template <typename W>
__global__ void compact_unpack_kernel(W *__restrict__ dst, const uint8_t *__restrict__ payload, size_t polys,
                                      uint32_t N, SerdeConsts sc, const LimbConst *__restrict__ limbs, uint32_t width) {
    const size_t idx = item_index();
    if (idx >= polys * N) return;
    const size_t poly = idx / N;
    const uint32_t i = static_cast<uint32_t>(idx % N);
    const int L = sc.limbs;
    if (width == 0) {
        for (int l = 0; l < L; ++l) dst[(poly * L + l) * N + i] = 0;
        return;
    }
    const size_t base = idx * static_cast<size_t>(width);
    const uint32_t mag_bits = width - 1;
    const size_t sbit = base + mag_bits;
    const bool neg = (payload[sbit >> 3] >> (sbit & 7)) & 1u;
    const uint32_t words = (mag_bits + 31) / 32, top_bits = mag_bits - (words - 1) * 32;  // top_bits in 1..32 (words >= 1)
    for (int l = 0; l < L; ++l) {
        const LimbConst lc = limbs[l];
        const uint64_t q = lc.q;
        uint64_t r = 0;
        for (uint32_t j = words; j-- > 0;) {
            uint32_t w = payload_bits32(payload, base + 32u * j);
            if (j == words - 1 && top_bits < 32) w &= (1u << top_bits) - 1u;
            if constexpr (sizeof(W) == 4) {
                const uint64_t x = (r << 32) | w;  // r < q < 2^31
                r = x - __umul64hi(x, lc.mu64) * q;
                if (r >= q) r -= q;
            } else {
                const u128_t x = (static_cast<u128_t>(r) << 32) | w;
                r = lc.kbits >= 32 ? barrett_reduce(x, q, lc.mu, lc.kbits) : static_cast<uint64_t>(x % q);
            }
        }
        if (mag_bits == 0) r = 0;
        if (neg && r) r = q - r;
        dst[(poly * L + l) * N + i] = static_cast<W>(r);
    }
}

static int build_consts(const GpuMatrix *mat, SerdeConsts &sc) {
    const GpuContext *ctx = mat->ctx;
    const int L = mat->level + 1;
    sc.limbs = L;
    // Q = product of the active moduli, little-endian 64-bit words
    std::vector<uint64_t> Q(1, 1);
    for (int l = 0; l < L; ++l) {
        sc.q[l] = ctx->moduli[l];
        unsigned __int128 carry = 0;
        for (size_t w = 0; w < Q.size(); ++w) {
            unsigned __int128 p = static_cast<unsigned __int128>(Q[w]) * ctx->moduli[l] + carry;
            Q[w] = static_cast<uint64_t>(p);
            carry = p >> 64;
        }
        if (carry) Q.push_back(static_cast<uint64_t>(carry));
    }
    if (Q.size() > static_cast<size_t>(kMaxWords)) return set_error("compact bytes: modulus exceeds 4096 bits");
    sc.words = static_cast<int>(Q.size());
    for (int w = 0; w < kMaxWords; ++w) {
        sc.modulus[w] = w < sc.words ? Q[w] : 0;
    }
    for (int w = 0; w < kMaxWords; ++w) {
        const uint64_t lo = sc.modulus[w] >> 1;
        const uint64_t hi = (w + 1 < kMaxWords) ? (sc.modulus[w + 1] & 1ull) << 63 : 0;
        sc.half[w] = lo | hi;
    }
    return 0;
}

extern "C" int gpu_matrix_store_compact_bytes(GpuMatrix *mat, uint8_t *payload_out, size_t payload_capacity,
                                              uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                              size_t *out_payload_len) {
    ABI_GUARD_BEGIN
    if (!mat || !out_max_coeff_bits || !out_bytes_per_coeff || !out_payload_len)
        return set_error("invalid gpu_matrix_store_compact_bytes arguments");
    *out_max_coeff_bits = 0;
    *out_bytes_per_coeff = 0;
    *out_payload_len = 0;
    GpuContext *ctx = mat->ctx;
    const size_t polys = matrix_polys(mat);
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    // the matrix is converted to COEFF in place (MatrixSerde.cu:1108-1118); the Rust side
    // records the original tag and re-NTTs on load
    if (mat->format == GPU_POLY_FORMAT_EVAL) {
        int rc = gpu_matrix_intt_all(mat);
        if (rc) return rc;
    }
    SerdeConsts sc;
    if (build_consts(mat, sc)) return 1;
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    const size_t coeffs = polys * N;
    const dim3 blocks = item_grid(coeffs, 256);
    const size_t gstride = static_cast<size_t>(ctx->limb_count);
    CtxBlock max_block(ctx);
    if (max_block.alloc(2 * sizeof(unsigned int))) return 1;  // [0] running maximum of the widths, [1] "needs the general path"
    void *const d_max = max_block.ptr;
    HIP_TRY(hipMemsetAsync(d_max, 0, 2 * sizeof(unsigned int), ctx->stream));
    // fast-path-only kernels first (up to 16 limbs): preimages, trapdoors and Gaussian-sized keys never leave them
    bool fast = sc.limbs <= 16 && !ctx->env.serde_general;
    unsigned int h_mf[2] = {0, 0};
    if (fast) {
        const dim3 fast_blocks(static_cast<unsigned>(std::min<size_t>((coeffs + 255) / 256, 8192)));
#define FAST_MAXBITS(WT, ML)                                                                                             \
    MXX_LAUNCH((compact_maxbits_fast_kernel<WT, ML>), fast_blocks, dim3(256), 0, ctx->stream, static_cast<const WT *>(mat->data), polys, N, sc, \
               ctx->d_garner, gstride, ctx->d_limbs, static_cast<unsigned int *>(d_max))
        if (ctx->wide) {
            if (sc.limbs <= 8) FAST_MAXBITS(uint64_t, 8);
            else FAST_MAXBITS(uint64_t, 16);
        } else {
            if (sc.limbs <= 8) FAST_MAXBITS(uint32_t, 8);
            else FAST_MAXBITS(uint32_t, 16);
        }
#undef FAST_MAXBITS
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_mf, d_max, sizeof(h_mf), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (h_mf[1]) {  // a coefficient beyond the fast paths: start over with the general kernels
            fast = false;
            HIP_TRY(hipMemsetAsync(d_max, 0, 2 * sizeof(unsigned int), ctx->stream));
        }
    }
#define SERDE_LAUNCH(KERNEL, WT, ...)                                                                                  \
    do {                                                                                                               \
        if (sc.limbs <= 8) MXX_LAUNCH((KERNEL<WT, 8>), blocks, dim3(256), 0, ctx->stream, __VA_ARGS__);         \
        else if (sc.limbs <= 16) MXX_LAUNCH((KERNEL<WT, 16>), blocks, dim3(256), 0, ctx->stream, __VA_ARGS__);  \
        else MXX_LAUNCH((KERNEL<WT, 64>), blocks, dim3(256), 0, ctx->stream, __VA_ARGS__);                      \
    } while (0)
    unsigned int h_max = h_mf[0];
    if (!fast) {
        if (ctx->wide)
            SERDE_LAUNCH(compact_maxbits_kernel, uint64_t, static_cast<const uint64_t *>(mat->data), polys, N, sc, ctx->d_garner,
                         gstride, ctx->d_limbs, static_cast<unsigned int *>(d_max));
        else
            SERDE_LAUNCH(compact_maxbits_kernel, uint32_t, static_cast<const uint32_t *>(mat->data), polys, N, sc, ctx->d_garner,
                         gstride, ctx->d_limbs, static_cast<unsigned int *>(d_max));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h_max, d_max, sizeof(h_max), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    const unsigned int width = h_max == 0 ? 0 : h_max + 1;
    if (width > 0xffffu) return set_error("centered max coeff bits exceed u16 range in gpu_matrix_store_compact_bytes");
    const unsigned int bytes_per_coeff = (width + 7) / 8;
    const size_t total_bits = coeffs * static_cast<size_t>(width);
    const size_t payload_len = (total_bits + 7) / 8;
    if (payload_len > payload_capacity) {
        // the caller learns the width and the length it needs (the reference reports only the error): a host that does not
        // want to pin the worst case - bits(Q) per coefficient - retries once with the exact size
        *out_max_coeff_bits = static_cast<uint16_t>(width);
        *out_bytes_per_coeff = static_cast<uint16_t>(bytes_per_coeff);
        *out_payload_len = payload_len;
        return set_error("payload buffer too small in gpu_matrix_store_compact_bytes");
    }
    if (payload_len > 0) {
        if (!payload_out) return set_error("null payload buffer in gpu_matrix_store_compact_bytes");
        const size_t padded = (payload_len + 3) / 4 * 4 + 4;
        CtxBlock payload_block(ctx);
        if (payload_block.alloc(padded)) return 1;
        void *const d_payload = payload_block.ptr;
        const bool in_lds = width <= kPackLdsWidth;
        if (!in_lds) HIP_TRY(hipMemsetAsync(d_payload, 0, padded, ctx->stream));
        const size_t pack_lds = in_lds ? 8u * width * sizeof(uint32_t) : 0;
        const size_t word_count = padded / 4;
#undef SERDE_LAUNCH
#define PACK_LAUNCH(WT, ML, LDSF)                                                                                       \
    MXX_LAUNCH((compact_pack_kernel<WT, ML, LDSF>), blocks, dim3(256), pack_lds, ctx->stream, static_cast<const WT *>(mat->data), polys, N, sc, \
               ctx->d_garner, gstride, ctx->d_limbs, width, static_cast<uint32_t *>(d_payload), word_count)
#define PACK_BY_LIMBS(WT, LDSF)                       \
    do {                                              \
        if (sc.limbs <= 8) PACK_LAUNCH(WT, 8, LDSF);  \
        else if (sc.limbs <= 16) PACK_LAUNCH(WT, 16, LDSF); \
        else PACK_LAUNCH(WT, 64, LDSF);               \
    } while (0)
#define FAST_PACK(WT, ML)                                                                                                \
    MXX_LAUNCH((compact_pack_fast_kernel<WT, ML>), blocks, dim3(256), pack_lds, ctx->stream, static_cast<const WT *>(mat->data), polys, N, sc, \
               ctx->d_garner, gstride, ctx->d_limbs, width, static_cast<uint32_t *>(d_payload), word_count)
        if (fast && in_lds) {
            if (ctx->wide) {
                if (sc.limbs <= 8) FAST_PACK(uint64_t, 8);
                else FAST_PACK(uint64_t, 16);
            } else {
                if (sc.limbs <= 8) FAST_PACK(uint32_t, 8);
                else FAST_PACK(uint32_t, 16);
            }
        } else if (ctx->wide) {
            if (in_lds) PACK_BY_LIMBS(uint64_t, true);
            else PACK_BY_LIMBS(uint64_t, false);
        } else {
            if (in_lds) PACK_BY_LIMBS(uint32_t, true);
            else PACK_BY_LIMBS(uint32_t, false);
        }
#undef FAST_PACK
#undef PACK_BY_LIMBS
#undef PACK_LAUNCH
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(payload_out, d_payload, payload_len, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *out_max_coeff_bits = static_cast<uint16_t>(width);
    *out_bytes_per_coeff = static_cast<uint16_t>(bytes_per_coeff);
    *out_payload_len = payload_len;
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_load_compact_bytes(GpuMatrix *mat, const uint8_t *payload, size_t payload_len,
                                             uint16_t max_coeff_bits) {
    ABI_GUARD_BEGIN
    if (!mat) return set_error("invalid gpu_matrix_load_compact_bytes arguments");
    GpuContext *ctx = mat->ctx;
    const size_t polys = matrix_polys(mat);
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    const size_t coeffs = polys * N;
    if (max_coeff_bits == 0) {
        if (payload_len != 0) return set_error("payload_len must be zero when max_coeff_bits is zero");
    } else if (!payload && coeffs) {
        return set_error("null payload in gpu_matrix_load_compact_bytes");
    }
    const size_t expected = (coeffs * static_cast<size_t>(max_coeff_bits) + 7) / 8;
    if (payload_len != expected) return set_error("payload length mismatch in gpu_matrix_load_compact_bytes");
    mat->format = GPU_POLY_FORMAT_COEFF;
    if (coeffs == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    SerdeConsts sc;
    if (build_consts(mat, sc)) return 1;
    CtxBlock payload_block(ctx);
    if (payload_len) {
        if (payload_block.alloc(payload_len + 8)) return 1;  // payload_bits32 reads up to 4 bytes past a coefficient
        HIP_TRY(hipMemsetAsync(static_cast<uint8_t *>(payload_block.ptr) + payload_len, 0, 8, ctx->stream));
        HIP_TRY(hipMemcpyAsync(payload_block.ptr, payload, payload_len, hipMemcpyHostToDevice, ctx->stream));
    }
    void *const d_payload = payload_block.ptr;
    const dim3 blocks = item_grid(coeffs, 256);
    if (ctx->wide)
        MXX_LAUNCH(compact_unpack_kernel<uint64_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(mat->data), static_cast<const uint8_t *>(d_payload), polys, N, sc,
                           ctx->d_limbs, static_cast<uint32_t>(max_coeff_bits));
    else
        MXX_LAUNCH(compact_unpack_kernel<uint32_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(mat->data), static_cast<const uint8_t *>(d_payload), polys, N, sc,
                           ctx->d_limbs, static_cast<uint32_t>(max_coeff_bits));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // synchronous, like the reference; payload may be freed by the caller
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_poly_store_compact_bytes(GpuMatrix *poly, uint8_t *payload_out, size_t payload_capacity,
                                            uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                            size_t *out_payload_len) {
    return gpu_matrix_store_compact_bytes(poly, payload_out, payload_capacity, out_max_coeff_bits, out_bytes_per_coeff,
                                          out_payload_len);
}

extern "C" int gpu_poly_load_compact_bytes(GpuMatrix *poly, const uint8_t *payload, size_t payload_len,
                                           uint16_t max_coeff_bits) {
    return gpu_matrix_load_compact_bytes(poly, payload, payload_len, max_coeff_bits);
}
