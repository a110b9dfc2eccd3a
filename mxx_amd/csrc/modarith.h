// modarith.h — device modular arithmetic for residues held in uint32_t (q < 2^31)
// or uint64_t (q < 2^62).  Everything stays in registers: Shoup multiplication for
// fixed multiplicands (twiddles, n^-1), Barrett for variable x variable products,
// mulhi-by-floor(2^64/q) for lazily accumulated 64-bit sums.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

typedef unsigned __int128 u128_t;

template <typename W>
struct Wide;
template <>
struct Wide<uint32_t> {
    typedef uint64_t type;
};
template <>
struct Wide<uint64_t> {
    typedef u128_t type;
};

__device__ __forceinline__ uint32_t mulhi_w(uint32_t a, uint32_t b) { return __umulhi(a, b); }
__device__ __forceinline__ uint64_t mulhi_w(uint64_t a, uint64_t b) { return __umul64hi(a, b); }

template <typename W>
__device__ __forceinline__ W add_mod(W a, W b, W q) {
    W r = a + b;
    return r >= q ? r - q : r;
}

template <typename W>
__device__ __forceinline__ W sub_mod(W a, W b, W q) {
    return a >= b ? a - b : a + q - b;
}

// x*w mod q with the Shoup companion wsh = floor(w * 2^bits(W) / q); any x in [0, 2^bits(W)).
template <typename W>
__device__ __forceinline__ W mul_shoup(W x, W w, W wsh, W q) {
    W t = mulhi_w(x, wsh);
    W r = x * w - t * q;  // in [0, 2q)
    return min(r, static_cast<W>(r - q));
}

// lazy variant: result in [0, 2q)
template <typename W>
__device__ __forceinline__ W mul_shoup_lazy(W x, W w, W wsh, W q) {
    W t = mulhi_w(x, wsh);
    return x * w - t * q;
}

// Barrett reduction of a double-width value x < 2^(2k), k = bits(q), mu = floor(2^(2k)/q).
__device__ __forceinline__ uint32_t barrett_reduce(uint64_t x, uint32_t q, uint64_t mu, uint32_t k) {
    uint64_t t = x >> (k - 1);            // < 2^(k+1) <= 2^32
    uint64_t qhat = (t * mu) >> (k + 1);  // t*mu < 2^(2k+2) <= 2^64
    uint64_t r = x - qhat * q;            // in [0, 3q)
    if (r >= q) r -= q;
    if (r >= q) r -= q;
    return static_cast<uint32_t>(r);
}

__device__ __forceinline__ uint64_t barrett_reduce(u128_t x, uint64_t q, uint64_t mu, uint32_t k) {
    uint64_t t = static_cast<uint64_t>(x >> (k - 1));  // < 2^(k+1) <= 2^63
    uint64_t qhat = static_cast<uint64_t>((static_cast<u128_t>(t) * mu) >> (k + 1));
    uint64_t r = static_cast<uint64_t>(x) - qhat * q;  // true value in [0, 3q) < 2^64
    if (r >= q) r -= q;
    if (r >= q) r -= q;
    return r;
}

template <typename W>
__device__ __forceinline__ W mul_mod(W a, W b, W q, uint64_t mu, uint32_t k) {
    typedef typename Wide<W>::type D;
    return barrett_reduce(static_cast<D>(a) * b, q, mu, k);
}

// acc < 2^64 (lazily accumulated), q < 2^32, mu64 = floor(2^64/q): result in [0,q)
__device__ __forceinline__ uint32_t reduce_u64_sum(uint64_t acc, uint32_t q, uint64_t mu64) {
    uint64_t qhat = __umul64hi(acc, mu64);
    uint64_t r = acc - qhat * q;  // in [0, 2q)
    if (r >= q) r -= q;
    return static_cast<uint32_t>(r);
}

// The same reduction when a bound on the sum is known: acc < 2^W with W >= 32, q < 2^30 and W - bits(q) <= 31.
// One 32-bit Barrett step on the top 32 of the W bits: t = acc >> (W - 32), qhat = hi32(t * floor(2^W / q)) is at most 3
// short of the quotient, so the remainder fits 32 bits (< 4q) and two conditional subtractions finish: 8 VALU
// instructions against ~17 for the 64 x 64 -> high-64 product above.  floor(2^W / q) = mu64 >> (64 - W).
struct BoundedReduce {
    uint32_t shift, mu;  // W - 32, floor(2^W / q)
    bool ok;
};
__device__ __forceinline__ BoundedReduce bounded_reduce_setup(uint32_t kbits, uint64_t mu64, uint32_t terms) {
    // sum of `terms` products of residues below 2^kbits
    const uint32_t lg = terms <= 1 ? 0 : 32 - __builtin_clz(terms - 1);
    uint32_t w = 2 * kbits + lg;
    if (w < 32) w = 32;
    BoundedReduce b;
    b.ok = kbits <= 29 && w <= 63 && w - kbits <= 31;
    b.shift = b.ok ? w - 32 : 0;
    b.mu = b.ok ? static_cast<uint32_t>(mu64 >> (64 - w)) : 0;
    return b;
}
__device__ __forceinline__ uint32_t reduce_u64_bounded(uint64_t acc, uint32_t q, const BoundedReduce &b) {
    const uint32_t t = static_cast<uint32_t>(acc >> b.shift);
    const uint32_t qhat = __umulhi(t, b.mu);
    uint32_t r = static_cast<uint32_t>(acc) - qhat * q;  // true remainder estimate, in [0, 4q)
    r = min(r, r - 2 * q);
    return min(r, r - q);
}

// acc < 2^128 (lazily accumulated products of 64-bit residues), q < 2^62: result in [0,q).
// acc = x1 2^64 + x0 = (x1 mod q) R + (x0 mod q) (mod q) with R = 2^64 mod q = -q mu64 (mod 2^64); the three
// Barrett steps need their inputs below 2^(2k), which 64-bit words are once k >= 33.
__device__ __forceinline__ uint64_t reduce_u128_sum(u128_t acc, uint64_t q, uint64_t mu, uint32_t k, uint64_t mu64) {
    if (k < 33) return static_cast<uint64_t>(acc % q);  // small prime in a wide context: rare, keep it simple
    const uint64_t x1 = static_cast<uint64_t>(acc >> 64), x0 = static_cast<uint64_t>(acc);
    const uint64_t y1 = barrett_reduce(static_cast<u128_t>(x1), q, mu, k);
    const uint64_t y0 = barrett_reduce(static_cast<u128_t>(x0), q, mu, k);
    const uint64_t r64 = 0ull - q * mu64;
    return barrett_reduce(static_cast<u128_t>(y1) * r64 + y0, q, mu, k);
}

// signed 64-bit integer -> residue in [0,q)
template <typename W>
__device__ __forceinline__ W signed_to_residue(int64_t v, W q) {
    if (v >= 0) return static_cast<W>(static_cast<uint64_t>(v) % q);
    uint64_t mag = static_cast<uint64_t>(-(v + 1)) + 1;
    uint64_t rem = mag % q;
    return rem == 0 ? static_cast<W>(0) : static_cast<W>(q - rem);
}

// same value as signed_to_residue without the 64-bit `%` (no hardware divider): mu64 = floor(2^64/q)
template <typename W>
__device__ __forceinline__ W signed_to_residue_mu(int64_t v, uint64_t q, uint64_t mu64) {
    const int64_t sq = static_cast<int64_t>(q);  // q < 2^62
    if (v > -sq && v < sq) return static_cast<W>(v < 0 ? v + sq : v);  // digits, small Gaussians
    const uint64_t mag = v >= 0 ? static_cast<uint64_t>(v) : static_cast<uint64_t>(-(v + 1)) + 1;
    uint64_t r;
    if ((mag >> 32) == 0)  // quotient estimate from the top word of mu64 alone
        r = mag - static_cast<uint64_t>(__umulhi(static_cast<uint32_t>(mag), static_cast<uint32_t>(mu64 >> 32))) * q;
    else
        r = mag - __umul64hi(mag, mu64) * q;
    while (r >= q) r -= q;  // either estimate is at most 3 short
    return static_cast<W>((v >= 0 || r == 0) ? r : q - r);
}

// residue -> centred representative in (-q/2, q/2]  (reference: cuda/src/matrix/MatrixSampling.cu:193-208)
__device__ __forceinline__ int64_t centered_residue(uint64_t v, uint64_t q) {
    uint64_t half = q >> 1;
    if (v <= half) return static_cast<int64_t>(v);
    return -static_cast<int64_t>(q - v);
}
