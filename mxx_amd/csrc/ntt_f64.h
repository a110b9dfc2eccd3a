// ntt_f64.h — whole-vector negacyclic NTT / INTT for 64-bit residue words whose moduli are below 2^51, computed in
// DOUBLE PRECISION (round 3).
//
// Why: with 64-bit words a lazy butterfly costs ~20 integer VALU instructions (a 64 x 64 high product is four 32-bit
// multiplies plus carries; two low products; the subtraction), against 5 for 32-bit words, and every one of them is
// priced ~4.5 SIMD cycles on gfx950 - the 51-bit transforms ran at 2.0-3.0 TB/s, VALU-bound.  An FP64 instruction costs
// the same as ONE 32-bit multiply.  Residues are exact integers in doubles (|x| < 2^53), and for a table constant w
//     c = rint(V * (w / q));  h = V * w;  l = fma(V, w, -h);  T = fma(-c, q, h) + l
// is V w - c q exactly: the FMA recovers the rounding error of the product, h - c q is an integer below 2^53, and so is
// the final sum.  |T| <= (1/2 + e/2) q for |V| <= e q (the quotient estimate errs by at most 1/2 + e/2).  Six
// instructions per product; the butterfly A = U + T, B = U - T adds two; values are folded (x - q rint(x / q), three
// instructions) when the next stage could pass 2^53 - every second stage for 51-bit moduli, once per pass below 2^49.
// tools/bfly_u64_f64.hip: 54 against 105 cycles per wave-butterfly, bit-exact against 128-bit arithmetic.
//
// Same transform, tables (psi = minimum primitive root, bit-reversed order) and three-pass LDS structure as the integer
// kernels of ntt_lds.h; global data stays uint64_t residues; outputs are canonical and bit-identical to the CPU oracle
// and to the integer kernels (tests/test_gpu_core.py compares all three).
#pragma once

#include "ntt_lds.h"

struct TwF {        // twiddle and twiddle / q
    double w, wi;
};
struct F64Limb {    // per limb: q, 1 / q, and the two constants of the last inverse stage with N^-1 folded in
    double q, qinv;
    TwF n_inv, last_w;
};

namespace nttf {
constexpr int kFolded = 2;  // bound after a fold, in units of q / 4 (|x| <= q / 2; rounding slack lives in ELIM)

__device__ __forceinline__ double mulmod(double v, const TwF t, double q) {
    const double c = rint(v * t.wi);
    const double h = v * t.w;
    const double l = fma(v, t.w, -h);
    return fma(-c, q, h) + l;
}
__device__ __forceinline__ double fold(double x, double q, double qinv) { return fma(-rint(x * qinv), q, x); }

// bound (units of q / 4) of both butterfly outputs when both inputs are at E
__host__ __device__ constexpr int fwd_next(int E) { return E + 2 + (E + 1) / 2; }  // U + T, |T| <= (1/2 + e/2) q
__host__ __device__ constexpr int inv_next(int E) { return 2 * E; }               // X + Y, X - Y; the product is smaller

// ---- forward: stages [S_P, S_P + C) on 2^C values, bound E on entry; folds inserted at compile time -----------------
template <int C, int K, int E, int ELIM>
struct CtStages {
    static __device__ __forceinline__ void run(double (&v)[1 << C], const TwF *__restrict__ tw, uint32_t bi, int s_p, double q, double qinv) {
        if constexpr (K < C) {
            constexpr bool need = fwd_next(E) > ELIM;
            constexpr int e_in = need ? kFolded : E;
            static_assert(fwd_next(e_in) <= ELIM, "a folded input must fit");
            if constexpr (need) {
#pragma unroll
                for (int u = 0; u < (1 << C); ++u) v[u] = fold(v[u], q, qinv);
            }
            constexpr int half = 1 << (C - K - 1);
            const uint32_t tb = (1u << (s_p + K)) + (bi << K);
#pragma unroll
            for (int u = 0; u < (1 << C); ++u) {
                if (u & half) continue;
                const TwF t = tw[tb + (static_cast<uint32_t>(u) >> (C - K))];
                const double T = mulmod(v[u + half], t, q);
                const double U = v[u];
                v[u] = U + T;
                v[u + half] = U - T;
            }
            CtStages<C, K + 1, fwd_next(e_in), ELIM>::run(v, tw, bi, s_p, q, qinv);
        }
    }
};

// ---- inverse: Gentleman-Sande stages K = C-1 .. 0; LAST: the final stage multiplies both outputs (N^-1 folded in) ---
template <int C, int K, int E, int ELIM, bool LAST>
struct GsStages {
    static __device__ __forceinline__ void run(double (&v)[1 << C], const TwF *__restrict__ tw, uint32_t bi, int s_p, const F64Limb &lc) {
        if constexpr (K >= 0) {
            constexpr bool need = inv_next(E) > ELIM;
            constexpr int e_in = need ? kFolded : E;
            if constexpr (need) {
#pragma unroll
                for (int u = 0; u < (1 << C); ++u) v[u] = fold(v[u], lc.q, lc.qinv);
            }
            constexpr int half = 1 << (C - K - 1);
            const uint32_t tb = (1u << (s_p + K)) + (bi << K);
#pragma unroll
            for (int u = 0; u < (1 << C); ++u) {
                if (u & half) continue;
                const double X = v[u], Y = v[u + half];
                const double A = X + Y, D = X - Y;
                if constexpr (LAST && K == 0) {
                    v[u] = mulmod(A, lc.n_inv, lc.q);
                    v[u + half] = mulmod(D, lc.last_w, lc.q);
                } else {
                    v[u] = A;
                    v[u + half] = mulmod(D, tw[tb + (static_cast<uint32_t>(u) >> (C - K))], lc.q);
                }
            }
            GsStages<C, K - 1, inv_next(e_in), ELIM, LAST>::run(v, tw, bi, s_p, lc);
        }
    }
};

__device__ __forceinline__ uint64_t to_residue(double x, double q, double qinv) {  // any bound the folds allow -> [0, q)
    double r = fold(x, q, qinv);  // [-q/2, q/2] up to the quotient's rounding
    r = r < 0.0 ? r + q : r;
    r = r >= q ? r - q : r;
    return static_cast<uint64_t>(r);
}

// what the split kernels hand over in the matrix's own 8-byte slots between their two launches: folded doubles
__device__ __forceinline__ double raw_f64(uint64_t bits) { return __longlong_as_double(static_cast<long long>(bits)); }
__device__ __forceinline__ uint64_t f64_raw(double x) { return static_cast<uint64_t>(__double_as_longlong(x)); }

// ---- small rings (2..512 points): one radix-2 stage per barrier, the vector as doubles in LDS ------------------------
// The rings of the GGH15 chain (n = 256, 51-bit limbs) used the integer kernel of ntt.hip - a 64 x 64 Shoup product per
// butterfly, ~30 VALU instructions; the same loop on doubles is mulmod's six + the butterfly's two (+ folds).  Bounds in
// units of q / 4 for the tightest case (51-bit moduli: 15 units, as ELIM of the whole-vector kernels): canonical inputs
// are at 4; forward 4 -> 8 -> 14, then a fold (2) before every second stage (2 -> 5 -> 10); inverse 4 -> 8, then a fold
// before every second stage (2 -> 4 -> 8) - the schedule is fixed, so smaller moduli just fold more often than they must.  Same tables and
// bit-reversed order as every other kernel; outputs canonical, bit-identical to the integer kernel.
template <bool INV>
__global__ void small_kernel(uint64_t *__restrict__ data, const TwF *__restrict__ tw_all, const F64Limb *__restrict__ limbs,
                             uint32_t L, uint32_t logN) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *x = reinterpret_cast<double *>(smem);
    const uint32_t N = 1u << logN, tid = threadIdx.x, T = blockDim.x;
    const size_t vec = blockIdx.x;
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const F64Limb lc = limbs[limb];
    const TwF *tw = tw_all + static_cast<size_t>(limb) * N;
    uint64_t *g = data + vec * N;
    for (uint32_t i = tid; i < N; i += T) x[i] = static_cast<double>(g[i]);
    __syncthreads();
    if (!INV) {
        uint32_t logt = logN - 1;
        for (uint32_t s = 0; s < logN; ++s, --logt) {
            const uint32_t m = 1u << s, t = 1u << logt;
            const bool fold_in = s >= 2 && (s & 1u) == 0;
            for (uint32_t b = tid; b < N / 2; b += T) {
                const uint32_t i = b >> logt, j = b & (t - 1);
                const uint32_t lo = (i << (logt + 1)) + j, hi = lo + t;
                double U = x[lo], V = x[hi];
                if (fold_in) {
                    U = fold(U, lc.q, lc.qinv);
                    V = fold(V, lc.q, lc.qinv);
                }
                const double Tm = mulmod(V, tw[m + i], lc.q);
                x[lo] = U + Tm;
                x[hi] = U - Tm;
            }
            __syncthreads();
        }
    } else {
        uint32_t logt = 0;
        for (uint32_t s = 0; s < logN; ++s, ++logt) {
            const uint32_t m = N >> (s + 1), t = 1u << logt;
            const bool fold_in = (s & 1u) != 0, last = s + 1 == logN;
            for (uint32_t b = tid; b < N / 2; b += T) {
                const uint32_t i = b >> logt, j = b & (t - 1);
                const uint32_t lo = (i << (logt + 1)) + j, hi = lo + t;
                double X = x[lo], Y = x[hi];
                if (fold_in) {
                    X = fold(X, lc.q, lc.qinv);
                    Y = fold(Y, lc.q, lc.qinv);
                }
                const double A = X + Y, D = X - Y;
                if (last) {  // N^-1 folded into the last stage's two products (m = 1, i = 0)
                    x[lo] = mulmod(A, lc.n_inv, lc.q);
                    x[hi] = mulmod(D, lc.last_w, lc.q);
                } else {
                    x[lo] = A;
                    x[hi] = mulmod(D, tw[m + i], lc.q);
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < N; i += T) g[i] = to_residue(x[i], lc.q, lc.qinv);
}

// the transform of one (sub-)vector: `load(e)` supplies residue e (natural order); canonical residues, bit-reversed, to g.
// PRE > 0 (rings beyond LDS, as ntt_lds.h): the vector has 2^(LOGN + PRE) points, head_kernel did its first PRE stages and
// left folded doubles; this workgroup transforms sub-vector `sub`, twiddles at stage PRE + s, block (sub << s) + b.
template <int LOGN, int LOGR, int PRE, int ELIM, bool NT, typename Load>
__device__ __forceinline__ void fwd_body(uint64_t *__restrict__ g, const Load load, const TwF *__restrict__ tw, const F64Limb &lc, uint32_t sub) {
    typedef NttLdsCfg<uint64_t, LOGN, LOGR, false> Cfg;
    constexpr uint32_t N = Cfg::N, T = Cfg::T;
    constexpr int P = Cfg::P, CLAST = Cfg::CLAST, R = 1 << LOGR;
    static_assert(P == 3, "three passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *x = reinterpret_cast<double *>(smem);
    const uint32_t tid = threadIdx.x;
    const double q = lc.q, qinv = lc.qinv;
    {   // pass 0: stages [0, LOGR) on elements tid + T u straight from HBM; inputs are residues (bound q: 4 units)
        double v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = PRE > 0 ? raw_f64(load(tid + T * u)) : static_cast<double>(load(tid + T * u));
        CtStages<LOGR, 0, (PRE > 0 ? kFolded : 4), ELIM>::run(v, tw, sub, PRE, q, qinv);
        const uint32_t pb = lds_pad_c(tid);
#pragma unroll
        for (int u = 0; u < R; ++u) x[pb + lds_pad_c(T * u)] = fold(v[u], q, qinv);  // every pass hands over folded values
    }
    __syncthreads();
    {   // pass 1: stages [LOGR, 2 LOGR)
        constexpr uint32_t B = 1u << (LOGN - LOGR), S = B >> LOGR;
        uint32_t bi, pb;
        lds_set_addr<uint64_t, LOGN, LOGR, LOGR, LOGR>(tid, 0, bi, pb);
        double v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = x[pb + lds_pad_c(S * u)];
        CtStages<LOGR, 0, kFolded, ELIM>::run(v, tw, (sub << LOGR) + bi, PRE + LOGR, q, qinv);
#pragma unroll
        for (int u = 0; u < R; ++u) x[pb + lds_pad_c(S * u)] = fold(v[u], q, qinv);
    }
    __syncthreads();
    uint64_t *xr = reinterpret_cast<uint64_t *>(smem);  // the last pass leaves canonical residues in place
    {   // pass 2: stages [2 LOGR, LOGN): R contiguous values per thread
        constexpr int G = 1 << (LOGR - CLAST), E = 1 << CLAST;
        const uint32_t pb0 = lds_pad_c(tid * R);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            double v[E];
#pragma unroll
            for (int u = 0; u < E; ++u) v[u] = x[pb0 + gi * E + u];
            CtStages<CLAST, 0, kFolded, ELIM>::run(v, tw, (sub << (2 * LOGR)) + tid * G + gi, PRE + 2 * LOGR, q, qinv);
#pragma unroll
            for (int u = 0; u < E; ++u) xr[pb0 + gi * E + u] = to_residue(v[u], q, qinv);
        }
    }
    __syncthreads();
    // LDS -> HBM, 16 bytes per lane
#pragma unroll
    for (uint32_t jj = 0; jj < N / 2 / T; ++jj) {
        const uint32_t i = tid + jj * T;
        nt_store16<NT, uint64_t>(g + static_cast<size_t>(i) * 2, &xr[lds_pad_c(i * 2)]);
    }
}

template <int LOGN, int LOGR, int WAVES_PER_EU, int ELIM, bool NT, int PRE = 0>
__global__ void __launch_bounds__(1 << (LOGN - LOGR), WAVES_PER_EU)
    fwd_kernel(uint64_t *__restrict__ data, const TwF *__restrict__ tw_all, const F64Limb *__restrict__ limbs, uint32_t L) {
    const size_t vec = blockIdx.x >> PRE;
    const uint32_t sub = blockIdx.x & ((1u << PRE) - 1u);
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const F64Limb lc = limbs[limb];
    uint64_t *g = data + (vec << (LOGN + PRE)) + (static_cast<size_t>(sub) << LOGN);
    fwd_body<LOGN, LOGR, PRE, ELIM, NT>(g, LoadData<uint64_t, NT>{g}, tw_all + (static_cast<size_t>(limb) << (LOGN + PRE)), lc, sub);
}

// rings beyond LDS, forward: stages [0, PRE) on the strided sets {j + (N >> PRE) u}; folded doubles go back into the
// vector's own slots for the sub-vector kernels.  DIGITS: the gadget decomposition in the load (grid and indexing of
// ntt_fwd_head_digits_kernel)
template <int PRE, int ELIM, bool DIGITS, bool REDUCE, bool NTS>
__global__ void __launch_bounds__(256)
    head_kernel(uint64_t *__restrict__ out, const uint64_t *__restrict__ coeff, const TwF *__restrict__ tw_all,
                const F64Limb *__restrict__ flimbs, const LimbConst *__restrict__ limbs, uint32_t L, uint32_t logN, uint32_t src_cols,
                uint32_t dpt, uint32_t base_bits, uint32_t k) {
    constexpr int R = 1 << PRE;
    const uint32_t S = (1u << logN) >> PRE;
    const uint32_t sets_blocks = S / blockDim.x;
    uint32_t limb, j;
    uint64_t *g;
    const uint64_t *src;
    uint32_t sh = 0;
    uint64_t mask = ~0ull, qred = 0;
    if constexpr (DIGITS) {
        const uint32_t sb = blockIdx.x % sets_blocks, rest = blockIdx.x / sets_blocks;
        limb = rest % L;
        const uint32_t col = rest / L;
        const uint32_t td = blockIdx.y, t = td / dpt, d = td - t * dpt;
        const size_t r = blockIdx.z;
        j = sb * blockDim.x + threadIdx.x;
        const uint32_t src_bits = limbs[t].kbits, shift = d * base_bits;
        mask = 0;
        if (shift < src_bits && shift < 64) {
            const uint32_t rem = src_bits - shift;
            const uint32_t db = base_bits < rem ? base_bits : rem;
            mask = db >= 64 ? ~0ull : ((1ull << db) - 1);
        }
        sh = shift < 64 ? shift : 0;
        qred = limbs[limb].q;
        src = coeff + (((r * src_cols + col) * L + t) << logN) + j;
        g = out + (((((r * k + td) * src_cols + col) * L) + limb) << logN) + j;
    } else {
        const size_t vec = blockIdx.x / sets_blocks;
        j = (blockIdx.x - static_cast<uint32_t>(vec) * sets_blocks) * blockDim.x + threadIdx.x;
        limb = static_cast<uint32_t>(vec % L);
        g = out + (vec << logN) + j;
        src = g;
    }
    const F64Limb lc = flimbs[limb];
    const TwF *tw = tw_all + (static_cast<size_t>(limb) << logN);
    double v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        uint64_t x = src[static_cast<size_t>(S) * u];
        if constexpr (DIGITS) {
            x = (x >> sh) & mask;
            if constexpr (REDUCE) x = x >= qred ? x % qred : x;
        }
        v[u] = static_cast<double>(x);
    }
    CtStages<PRE, 0, 4, ELIM>::run(v, tw, 0, 0, lc.q, lc.qinv);
#pragma unroll
    for (int u = 0; u < R; ++u) nt_store<NTS, uint64_t>(f64_raw(fold(v[u], lc.q, lc.qinv)), g + static_cast<size_t>(S) * u);
}

// decompose + forward transform (decompose.hip): grid = (L * src_cols, k, source rows), as ntt_fwd_lazy_digits_kernel
template <int LOGN, int LOGR, int WAVES_PER_EU, int ELIM, bool REDUCE, bool NTS>
__global__ void __launch_bounds__(1 << (LOGN - LOGR), WAVES_PER_EU)
    fwd_digits_kernel(uint64_t *__restrict__ out, const uint64_t *__restrict__ coeff, const TwF *__restrict__ tw_all,
                      const F64Limb *__restrict__ flimbs, const LimbConst *__restrict__ limbs, uint32_t L, uint32_t src_cols,
                      uint32_t dpt, uint32_t base_bits, uint32_t k) {
    const uint32_t limb = blockIdx.x % L, col = blockIdx.x / L;
    const uint32_t td = blockIdx.y, t = td / dpt, d = td - t * dpt;
    const size_t r = blockIdx.z;
    const F64Limb lc = flimbs[limb];
    const uint64_t *src = coeff + (((r * src_cols + col) * L + t) << LOGN);
    uint64_t *g = out + (((((r * k + td) * src_cols + col) * L) + limb) << LOGN);
    fwd_body<LOGN, LOGR, 0, ELIM, NTS>(g, load_digit_of<uint64_t, REDUCE>(src, limbs, t, d, base_bits, limbs[limb].q),
                                       tw_all + (static_cast<size_t>(limb) << LOGN), lc, 0u);
}

// PRE > 0: the last PRE stages and the N^-1 scaling are tail_kernel's; this kernel then leaves folded doubles
template <int LOGN, int LOGR, int WAVES_PER_EU, int ELIM, bool NT, int PRE = 0>
__global__ void __launch_bounds__(1 << (LOGN - LOGR), WAVES_PER_EU)
    inv_kernel(uint64_t *__restrict__ data, const TwF *__restrict__ tw_all, const F64Limb *__restrict__ limbs, uint32_t L) {
    typedef NttLdsCfg<uint64_t, LOGN, LOGR, true> Cfg;
    constexpr uint32_t N = Cfg::N, T = Cfg::T;
    constexpr int P = Cfg::P, CLAST = Cfg::CLAST, R = 1 << LOGR;
    static_assert(P == 3, "three passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t *xr = reinterpret_cast<uint64_t *>(smem);
    double *x = reinterpret_cast<double *>(smem);
    const uint32_t tid = threadIdx.x;
    const size_t vec = blockIdx.x >> PRE;
    const uint32_t sub = blockIdx.x & ((1u << PRE) - 1u);
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const F64Limb lc = limbs[limb];
    const TwF *tw = tw_all + (static_cast<size_t>(limb) << (LOGN + PRE));
    uint64_t *g = data + (vec << (LOGN + PRE)) + (static_cast<size_t>(sub) << LOGN);
    // HBM -> LDS, 16 bytes per lane (residues as they are; converted by the first pass)
#pragma unroll
    for (uint32_t jj = 0; jj < N / 2 / T; ++jj) {
        const uint32_t i = tid + jj * T;
        nt_load16<NT, uint64_t>(&xr[lds_pad_c(i * 2)], g + static_cast<size_t>(i) * 2);
    }
    __syncthreads();
    {   // contiguous pass: stages [2 LOGR, LOGN) in GS order
        constexpr int G = 1 << (LOGR - CLAST), E = 1 << CLAST;
        const uint32_t pb0 = lds_pad_c(tid * R);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            double v[E];
#pragma unroll
            for (int u = 0; u < E; ++u) v[u] = static_cast<double>(xr[pb0 + gi * E + u]);
            GsStages<CLAST, CLAST - 1, 4, ELIM, false>::run(v, tw, (sub << (2 * LOGR)) + tid * G + gi, PRE + 2 * LOGR, lc);
#pragma unroll
            for (int u = 0; u < E; ++u) x[pb0 + gi * E + u] = fold(v[u], lc.q, lc.qinv);
        }
    }
    __syncthreads();
    {   // middle pass: stages [LOGR, 2 LOGR)
        constexpr uint32_t B = 1u << (LOGN - LOGR), S = B >> LOGR;
        uint32_t bi, pb;
        lds_set_addr<uint64_t, LOGN, LOGR, LOGR, LOGR>(tid, 0, bi, pb);
        double v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = x[pb + lds_pad_c(S * u)];
        GsStages<LOGR, LOGR - 1, kFolded, ELIM, false>::run(v, tw, (sub << LOGR) + bi, PRE + LOGR, lc);
#pragma unroll
        for (int u = 0; u < R; ++u) x[pb + lds_pad_c(S * u)] = fold(v[u], lc.q, lc.qinv);
    }
    __syncthreads();
    {   // strided pass: stages [0, LOGR), N^-1 in the last one, results straight to HBM (coalesced per u)
        const uint32_t pb = lds_pad_c(tid);
        double v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = x[pb + lds_pad_c(T * u)];
        if constexpr (PRE == 0) {
            GsStages<LOGR, LOGR - 1, kFolded, ELIM, true>::run(v, tw, 0, 0, lc);
#pragma unroll
            for (int u = 0; u < R; ++u) nt_store<NT, uint64_t>(to_residue(v[u], lc.q, lc.qinv), g + tid + T * u);
        } else {
            GsStages<LOGR, LOGR - 1, kFolded, ELIM, false>::run(v, tw, sub, PRE, lc);
#pragma unroll
            for (int u = 0; u < R; ++u) nt_store<NT, uint64_t>(f64_raw(fold(v[u], lc.q, lc.qinv)), g + tid + T * u);
        }
    }
}

// rings beyond LDS, inverse: the last PRE stages on the strided sets, N^-1 folded into the final one, canonical residues out
template <int PRE, int ELIM>
__global__ void __launch_bounds__(256)
    tail_kernel(uint64_t *__restrict__ data, const TwF *__restrict__ tw_all, const F64Limb *__restrict__ limbs, uint32_t L, uint32_t logN) {
    constexpr int R = 1 << PRE;
    const uint32_t S = (1u << logN) >> PRE;
    const uint32_t sets_blocks = S / blockDim.x;
    const size_t vec = blockIdx.x / sets_blocks;
    const uint32_t j = (blockIdx.x - static_cast<uint32_t>(vec) * sets_blocks) * blockDim.x + threadIdx.x;
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const F64Limb lc = limbs[limb];
    const TwF *tw = tw_all + (static_cast<size_t>(limb) << logN);
    uint64_t *g = data + (vec << logN) + j;
    double v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = raw_f64(g[static_cast<size_t>(S) * u]);
    GsStages<PRE, PRE - 1, kFolded, ELIM, true>::run(v, tw, 0, 0, lc);
#pragma unroll
    for (int u = 0; u < R; ++u) g[static_cast<size_t>(S) * u] = to_residue(v[u], lc.q, lc.qinv);
}
}  // namespace nttf
