// ntt_lds.h — the tuned, LDS-resident negacyclic NTT / INTT kernels (one workgroup per
// (polynomial, RNS limb) vector), with lazy (Harvey-style) butterflies.
//
// A 2^14-point transform is 114 688 butterflies per vector; on gfx950 a wave64 32-bit
// integer multiply (v_mul_lo/hi_u32, v_mad_u64_u32) costs ~3.7-4 SIMD cycles and an add ~2
// (tools/bfly_rates.hip: 15-19 cycles per 5-instruction butterfly per wave), which makes the
// kernel VALU-bound, not HBM-bound: the arithmetic alone takes longer than the 80 us the
// loads, LDS passes, barriers and stores take with the butterflies removed.  Every VALU
// instruction removed counts, so the butterflies keep values in redundant form:
//   forward (Cooley-Tukey):  nT = V*(-w) + hi(V*w')*q   (== -(V*w mod q), |.| < 2q)
//                            A = U - nT ; B = U + 2q + nT            -> 5 VALU ops
//       bounds only grow by 2q per stage, so no correction until the very end
//       ((1 + 2 logN) q < 2^W needs q < 2^(W-5)).
//   inverse (Gentleman-Sande): A = X + Y ; D = X + M - Y ; B = D*w - hi(D*w')*q  -> 6 ops
//       with M = the compile-time bound of Y; A doubles its bound per stage, so each
//       pass (<= 5 stages) ends with one mulhi-based fold back to [0, 2q) of the
//       elements that need it (2^(c+1) q < 2^W needs q < 2^(W-6)).
//   N^-1 is folded into the last inverse stage's twiddle.
// Same transform as ntt_generic_kernel (OpenFHE convention, see ntt.hip); outputs are
// canonical residues, bit-identical to the CPU oracle.
#pragma once

#include "common.h"
#include "modarith.h"

template <typename W>
struct TwPair {  // twiddle (negated for the forward transform) + Shoup companion, one load
    W w;
    W ws;
};

// LDS index padding (in words): +4 per 32, +16 per 512.
__host__ __device__ constexpr uint32_t lds_pad_c(uint32_t e) { return e + ((e >> 5) << 2) + ((e >> 9) << 4); }
static inline size_t lds_padded_words(size_t n) { return n + ((n >> 5) << 2) + ((n >> 9) << 4) + 16; }

template <typename W>
__device__ __forceinline__ W csub(W x, W m) {  // x in [0, 2m) -> [0, m)
    return min(x, static_cast<W>(x - m));
}

// Non-temporal accesses of the data vectors (template flag NT of the LDS kernels): a vector is read once and written once
// per kernel while the twiddle tables are re-read by every workgroup - the hint keeps the stream from displacing them in
// L2.  Same-box A/B (tools/sweep_ntt.py, 4096 polys x 4 limbs): helps the whole-vector-in-LDS kernels from 2^13 points
// (2^15 inverse 1.63 -> 1.46 ms, forward 1.26 -> 1.16; 64-bit words from 2^12: 2^14 inverse 2.10 -> 1.86), costs 6-14 % at
// 2^12 with 32-bit words and 5-9 % in the head / tail + sub-vector pairs, whose hand-over lives in L2 / the Infinity Cache.
// And it is only right for batches well beyond the Infinity Cache: whatever consumes a transform's output next otherwise
// finds it there (a 268 MB batch: the inverse behind a forward transform with non-temporal stores ran 123 -> 136 us).
// So: eligible kernels (ntt_nt_data) x batches of at least 1 GiB (the launcher).
template <typename W, int LOGN, int PRE>
constexpr bool ntt_nt_data() { return PRE == 0 && (sizeof(W) == 4 ? LOGN >= 13 : LOGN >= 12); }
template <bool NT, typename W>
__device__ __forceinline__ W nt_load(const W *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename W>
__device__ __forceinline__ void nt_store(W v, W *p) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT, typename W>
__device__ __forceinline__ void nt_load16(W *dst, const W *g) {  // 16 bytes global -> dst (LDS or registers)
    typedef W wx __attribute__((ext_vector_type(16 / sizeof(W))));
    if constexpr (NT) *reinterpret_cast<wx *>(dst) = __builtin_nontemporal_load(reinterpret_cast<const wx *>(g));
    else *reinterpret_cast<wx *>(dst) = *reinterpret_cast<const wx *>(g);
}
template <bool NT, typename W>
__device__ __forceinline__ void nt_store16(W *g, const W *src) {
    typedef W wx __attribute__((ext_vector_type(16 / sizeof(W))));
    if constexpr (NT) __builtin_nontemporal_store(*reinterpret_cast<const wx *>(src), reinterpret_cast<wx *>(g));
    else *reinterpret_cast<wx *>(g) = *reinterpret_cast<const wx *>(src);
}

// -q as a value the optimiser cannot see through: "x + t * nq" then stays one multiply-add (v_mad_u64_u32, low
// word) instead of being canonicalised back to a multiply and a subtract - same bits, one VALU instruction less
// per lazy product (one register).
template <typename W>
__device__ __forceinline__ W opaque_neg(W q) {
    W n = static_cast<W>(0) - q;
    asm("" : "+v"(n));
    return n;
}

// x < 2^bits(W) -> [0, 2q) with muw = floor(2^bits(W) / q)
template <typename W>
__device__ __forceinline__ W fold_2q(W x, W q, W muw) {
    return x + mulhi_w(x, muw) * opaque_neg<W>(q);
}


// ---- Montgomery product for a transform whose load multiplies by a resident ring element (MULW forms) ---------------
// REDC(a b) = a b 2^-32 (mod q) in [0, 2q): three multiply-class instructions, like a Shoup product, but on the plain
// 4-byte residue of the multiplier (no companion table).  The kernels that use it read the N^-1 constants from
// ctx->d_limbs_r, where they carry the compensating factor 2^32.
template <typename W>
__device__ __forceinline__ W neg_inv_pow2(W q) {  // -q^-1 mod 2^bits(W), q odd: Newton, 3 -> 6 -> 12 -> 24 -> 48 -> 96 bits
    W x = q;
#pragma unroll
    for (int i = 0; i < (sizeof(W) == 4 ? 4 : 5); ++i) x *= static_cast<W>(2) - q * x;
    return static_cast<W>(0) - x;
}
__device__ __forceinline__ uint32_t mont_mul_lazy(uint32_t a, uint32_t b, uint32_t q, uint32_t qninv) {
    const uint64_t p = static_cast<uint64_t>(a) * b;          // a b < q 2^32
    const uint32_t m = static_cast<uint32_t>(p) * qninv;
    return static_cast<uint32_t>((p + static_cast<uint64_t>(m) * q) >> 32);
}

// ---- forward pass: stages [S_P, S_P + C), Cooley-Tukey, values grow by 2q per stage --------
// NOTW (phase-timing builds only, ntt14.h): every twiddle is a register constant - the arithmetic without its table loads
template <typename W, int C, bool NOTW = false>
__device__ __forceinline__ void ct_network_lazy(W (&v)[1 << C], const TwPair<W> *__restrict__ tw, uint32_t bi, int s_p,
                                                W q, W twoq) {
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int half = 1 << (C - k - 1);
        const uint32_t tb = (1u << (s_p + k)) + (bi << k);
#pragma unroll
        for (int u = 0; u < (1 << C); ++u) {
            if (u & half) continue;
            const TwPair<W> t = NOTW ? TwPair<W>{static_cast<W>(q - 5u - tb), static_cast<W>(77u + tb)} : tw[tb + (static_cast<uint32_t>(u) >> (C - k))];
            const W V = v[u + half];
            const W nT = V * t.w + mulhi_w(V, t.ws) * q;  // t.w holds -w
            const W U = v[u];
            v[u] = U - nT;
            v[u + half] = U + twoq + nT;
        }
    }
}

// ---- TIGHT mode: moduli of 26..28 bits in 32-bit words (16 q <= 2^32) -----------------------------------------------
// The reference's end-to-end parameter sets use 28-bit limbs at n = 2^16 (tests/test_gpu_diamond_io.rs:64-70), past the
// "never correct" bound above.  The forward passes then start by bringing their inputs (below 16 q) under 8 q (passes
// of up to 4 stages, one subtract + min per element) or under 4 q (5 stages, two), so a pass still ends below 16 q;
// the inverse passes cap the bound exponents at kTightCap = 4 (see gs_exp_after).  Same bits out.
constexpr int kTightCap = 4;
template <typename W, int C, bool TIGHT>
__device__ __forceinline__ void ct_prefold(W (&v)[1 << C], W q) {
    if constexpr (TIGHT) {
#pragma unroll
        for (int u = 0; u < (1 << C); ++u) {
            v[u] = csub<W>(v[u], q << 3);
            if constexpr (C > 4) v[u] = csub<W>(v[u], q << 2);
        }
    }
}

// ---- inverse pass: Gentleman-Sande, input bound 2q, A-path bounds double per stage ----------
// bound exponent of element u after the stages that used bits 0..j (bound = 2^e * q).  A stage whose inputs sit at
// exponent e computes X + M - Y and X + Y below 2^(e+1) q; where that would pass 2^cap q the stage first folds both
// inputs back to [0, 2q) (exponent 1).  With the default cap no pass of up to 6 stages ever folds; with kTightCap
// two of the sixteen elements of a 4-stage pass do, before its last stage.
__host__ __device__ constexpr int gs_exp_after(int u, int j, int cap = 31) {
    int e = 1;
    for (int i = 0; i <= j; ++i) {
        if ((u >> i) & 1) {
            e = 1;
        } else {
            if (e + 1 > cap) e = 1;
            e = e + 1;
        }
    }
    return e;
}

// one stage (bit HB of the element index) of the pass; a function template per stage, because the optimiser gives up
// unrolling the stage loop of the 32-element passes for some instantiations (64-bit words; 2^13 and 2^15 points) -
// the element array then lives in scratch memory and the inverse transform runs 2-5x slower than the forward one
template <typename W, int C, bool LAST, int K, int CAP>
__device__ __forceinline__ void gs_stage_lazy(W (&v)[1 << C], const TwPair<W> *__restrict__ tw, uint32_t bi, int s_p, W q,
                                              const LimbConst &lc) {
    constexpr int hb = C - K - 1;  // bit used by this stage
    constexpr int half = 1 << hb;
    const uint32_t tb = (1u << (s_p + K)) + (bi << K);
#pragma unroll
    for (int u = 0; u < (1 << C); ++u) {
        if (u & half) continue;
        const int e_prev = hb == 0 ? 1 : gs_exp_after(u, hb - 1, CAP);  // same for u and u+half
        const bool pre = e_prev + 1 > CAP;
        const int e_in = pre ? 1 : e_prev;
        const W M = q << e_in;                                  // bound of Y
        W X = v[u], Y = v[u + half];
        if (pre) {
            const W muw = static_cast<W>(sizeof(W) == 4 ? lc.mu32 : lc.mu64);
            X = fold_2q<W>(X, q, muw);
            Y = fold_2q<W>(Y, q, muw);
        }
        const W D = X + M - Y;
        if (LAST && K == 0) {
            // last stage of the whole transform: fold N^-1 into both outputs
            const W A = X + Y, nq = opaque_neg<W>(q);
            v[u] = A * static_cast<W>(lc.n_inv) + mulhi_w(A, static_cast<W>(lc.n_inv_sh)) * nq;
            v[u + half] = D * static_cast<W>(lc.inv_last_w) + mulhi_w(D, static_cast<W>(lc.inv_last_w_sh)) * nq;
        } else {
            const TwPair<W> t = tw[tb + (static_cast<uint32_t>(u) >> (C - K))];
            v[u] = X + Y;
            v[u + half] = D * t.w + mulhi_w(D, t.ws) * opaque_neg<W>(q);  // [0, 2q); multiply-add form
        }
    }
}

template <typename W, int C, bool LAST, int K, int CAP>
struct GsStages {
    static __device__ __forceinline__ void run(W (&v)[1 << C], const TwPair<W> *__restrict__ tw, uint32_t bi, int s_p, W q,
                                               const LimbConst &lc) {
        gs_stage_lazy<W, C, LAST, K, CAP>(v, tw, bi, s_p, q, lc);
        if constexpr (K > 0) GsStages<W, C, LAST, K - 1, CAP>::run(v, tw, bi, s_p, q, lc);
    }
};

template <typename W, int C, bool LAST, int CAP = 31>
__device__ __forceinline__ void gs_network_lazy(W (&v)[1 << C], const TwPair<W> *__restrict__ tw, uint32_t bi, int s_p,
                                                W q, const LimbConst &lc) {
    GsStages<W, C, LAST, C - 1, CAP>::run(v, tw, bi, s_p, q, lc);  // stages K = C-1 .. 0
}

// bring every element of a finished inverse pass back to [0, 2q)
template <typename W, int C, int CAP = 31>
__device__ __forceinline__ void gs_fold(W (&v)[1 << C], W q, W muw) {
#pragma unroll
    for (int u = 0; u < (1 << C); ++u) {
        const int e = gs_exp_after(u, C - 1, CAP);
        if (e == 2) v[u] = csub<W>(v[u], q + q);
        else if (e > 2) v[u] = fold_2q<W>(v[u], q, muw);
    }
}

// ---- signed lazy butterflies for the inverse transform (u32 words, q < 2^24; used by ntt14.h) -----
// Values are int32 in redundant form.  With the centred twiddle w^ in (-q/2, q/2] and
// w^' = floor(w^ 2^32 / q), T = D w^ - mulhi_i32(D, w^') q equals D w (mod q) and lies in
// (-q |D| / 2^32, q + q |D| / 2^32): (-q/4, 5q/4) for |D| < 2^30.  The Gentleman-Sande butterfly becomes
// A = X + Y, D = X - Y (no +M offset to keep the difference non-negative: one VALU op less per butterfly)
// and B = T.  The A path doubles per stage; folds (x - mulhi_i32(x, floor(2^32/q)) q, same output range)
// keep every butterfly input below 2^5 q, so |X - Y| < 2^30.  Measured on the 2^14 inverse: 1970 -> 1726
// VALU instructions per thread, 156 -> ~138 us for M1.  The same idea in the forward transform
// (A = U + T, B = U - T instead of the 3-operand B = U + 2q + nT) is slower in the kernel (148 vs 137 us)
// although it wins in isolation (tools/bfly_forms.hip): the compiler splits the v_mad_u64_u32 into
// v_mul_lo + v_sub once the product is subtracted, and instruction count matters more than operand count.
__device__ __forceinline__ uint32_t smul_lazy(uint32_t x, uint32_t w, uint32_t ws, uint32_t q) {
    // x w + t (-q) (mod 2^32): v_mul_lo + v_mul_hi_i32 + v_mad_u64_u32 instead of two v_mul_lo and a v_sub
    return x * w + static_cast<uint32_t>(__mulhi(static_cast<int32_t>(x), static_cast<int32_t>(ws))) * opaque_neg<uint32_t>(q);
}
// bound exponent (|x| < 2^e q) of element u after the stages that used bits 0..j, all inputs at e0
__host__ __device__ constexpr int gs_exp_from(int e0, int u, int j) {
    int e = e0;
    for (int i = 0; i <= j; ++i) e = ((u >> i) & 1) ? 1 : e + 1;
    return e;
}

template <int C, bool LAST>
__device__ __forceinline__ void gs_network_signed(uint32_t (&v)[1 << C], const TwPair<uint32_t> *__restrict__ tw,
                                                  uint32_t bi, int s_p, uint32_t q, const LimbConst &lc) {
#pragma unroll
    for (int k = C - 1; k >= 0; --k) {
        const int half = 1 << (C - k - 1);
        const uint32_t tb = (1u << (s_p + k)) + (bi << k);
#pragma unroll
        for (int u = 0; u < (1 << C); ++u) {
            if (u & half) continue;
            const uint32_t X = v[u], Y = v[u + half];
            const uint32_t A = X + Y, D = X - Y;
            if (LAST && k == 0) {
                // last stage of the whole transform: |A|, |D| < 2^6 q; shift them to (0, 2^7 q) and
                // finish with the unsigned Shoup product by N^-1 (resp. w N^-1): outputs in [0, 2q)
                const uint32_t Ap = A + (q << 6), Dp = D + (q << 6);
                v[u] = Ap * static_cast<uint32_t>(lc.n_inv) + __umulhi(Ap, static_cast<uint32_t>(lc.n_inv_sh)) * opaque_neg<uint32_t>(q);
                v[u + half] = Dp * static_cast<uint32_t>(lc.inv_last_w) + __umulhi(Dp, static_cast<uint32_t>(lc.inv_last_w_sh)) * opaque_neg<uint32_t>(q);
            } else {
                const TwPair<uint32_t> t = tw[tb + (static_cast<uint32_t>(u) >> (C - k))];
                v[u] = A;
                v[u + half] = smul_lazy(D, t.w, t.ws, q);
            }
        }
    }
}

// after a pass whose inputs were all at exponent E0: fold the elements above KEEP back to exponent 1
template <int C, int E0, int KEEP>
__device__ __forceinline__ void gs_fold_signed(uint32_t (&v)[1 << C], uint32_t q, uint32_t mu32) {
#pragma unroll
    for (int u = 0; u < (1 << C); ++u)
        if (gs_exp_from(E0, u, C - 1) > KEEP)
            v[u] = v[u] + static_cast<uint32_t>(__mulhi(static_cast<int32_t>(v[u]), static_cast<int32_t>(mu32))) * opaque_neg<uint32_t>(q);
}

template <typename W, int LOGN, int LOGR, bool INV>
struct NttLdsCfg {
    static constexpr uint32_t N = 1u << LOGN;
    static constexpr uint32_t T = 1u << (LOGN - LOGR);
    static constexpr int P = (LOGN + LOGR - 1) / LOGR;
    static constexpr int CLAST = LOGN - (P - 1) * LOGR;
};

// strided first/last pass straight from/to HBM: element set {tid + T*u}
// middle and contiguous passes on LDS with compile-time offsets.
template <typename W, int LOGN, int LOGR, int S_P, int C>
__device__ __forceinline__ void lds_set_addr(uint32_t tid, int g, uint32_t &bi, uint32_t &pbase) {
    constexpr uint32_t B = 1u << (LOGN - S_P);
    constexpr uint32_t S = B >> C;
    constexpr int G = 1 << (LOGR - C);
    const uint32_t sigma = tid * G + g;
    bi = sigma / S;
    const uint32_t r = sigma - bi * S;
    pbase = lds_pad_c(bi * B + r);
}

// PRE > 0: the vector has 2^(LOGN + PRE) points and its first PRE stages were done by ntt_fwd_head_kernel; what is
// left are 2^PRE independent 2^LOGN-point sub-transforms (one workgroup each) whose twiddles sit at stage PRE + s,
// block (sub << s) + b of the full ring's table.
// the transform of one (sub-)vector: `load(e)` supplies element e of it, the result goes to g (NT: non-temporal stores)
template <typename W, int LOGN, int LOGR, int PRE, bool TIGHT, bool NT, typename Load>
__device__ __forceinline__ void ntt_fwd_lazy_body(W *__restrict__ g, const Load load, const TwPair<W> *__restrict__ tw,
                                                  const LimbConst &lc, uint32_t sub) {
    typedef NttLdsCfg<W, LOGN, LOGR, false> Cfg;
    constexpr uint32_t N = Cfg::N, T = Cfg::T;
    constexpr int P = Cfg::P, CLAST = Cfg::CLAST, R = 1 << LOGR;
    static_assert(P == 3, "three passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W *x = reinterpret_cast<W *>(smem);
    const uint32_t tid = threadIdx.x;
    const W q = static_cast<W>(lc.q), twoq = q + q;
    {   // pass 0: stages [0, LOGR), elements tid + T*u straight from HBM (coalesced per u)
        W v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = load(tid + T * u);
        if constexpr (PRE > 0) ct_prefold<W, LOGR, TIGHT>(v, q);  // the head kernel leaves (1 + 2 PRE) q
        ct_network_lazy<W, LOGR>(v, tw, sub, PRE, q, twoq);
        const uint32_t pb = lds_pad_c(tid);
#pragma unroll
        for (int u = 0; u < R; ++u) x[pb + lds_pad_c(T * u)] = v[u];
    }
    __syncthreads();
    {   // pass 1: stages [LOGR, 2 LOGR)
        constexpr uint32_t B = 1u << (LOGN - LOGR), S = B >> LOGR;
        uint32_t bi, pb;
        lds_set_addr<W, LOGN, LOGR, LOGR, LOGR>(tid, 0, bi, pb);
        W v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = x[pb + lds_pad_c(S * u)];
        ct_prefold<W, LOGR, TIGHT>(v, q);
        ct_network_lazy<W, LOGR>(v, tw, (sub << LOGR) + bi, PRE + LOGR, q, twoq);
#pragma unroll
        for (int u = 0; u < R; ++u) x[pb + lds_pad_c(S * u)] = v[u];
    }
    __syncthreads();
    {   // pass 2: stages [2 LOGR, LOGN): R contiguous words per thread
        constexpr int G = 1 << (LOGR - CLAST), E = 1 << CLAST;
        const uint32_t pb0 = lds_pad_c(tid * R);
        const W muw = static_cast<W>(sizeof(W) == 4 ? lc.mu32 : lc.mu64);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            W v[E];
#pragma unroll
            for (int u = 0; u < E; ++u) v[u] = x[pb0 + gi * E + u];
            ct_prefold<W, CLAST, TIGHT>(v, q);
            ct_network_lazy<W, CLAST>(v, tw, (sub << (2 * LOGR)) + tid * G + gi, PRE + 2 * LOGR, q, twoq);
            // canonical form: values < (1 + 2 logN) q
#pragma unroll
            for (int u = 0; u < E; ++u) x[pb0 + gi * E + u] = csub<W>(fold_2q<W>(v[u], q, muw), q);
        }
    }
    __syncthreads();
    // LDS -> HBM, 16 bytes per lane
    constexpr int VN = 16 / sizeof(W);
#pragma unroll
    for (uint32_t jj = 0; jj < N / VN / T; ++jj) {
        const uint32_t i = tid + jj * T;
        nt_store16<NT, W>(g + static_cast<size_t>(i) * VN, &x[lds_pad_c(i * VN)]);
    }
}


template <typename W, bool NT>
struct LoadData {
    const W *g;
    __device__ __forceinline__ W operator()(uint32_t e) const { return nt_load<NT, W>(g + e); }
};

template <typename W, int LOGN, int LOGR, int WAVES_PER_EU, int PRE = 0, bool TIGHT = false, bool NT = false>
__global__ void __launch_bounds__(1 << (LOGN - LOGR), WAVES_PER_EU)
    ntt_fwd_lazy_kernel(W *__restrict__ data, const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs,
                        uint32_t L) {
    const size_t vec = blockIdx.x >> PRE;
    const uint32_t sub = blockIdx.x & ((1u << PRE) - 1u);
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const LimbConst lc = limbs[limb];
    const TwPair<W> *tw = tw_all + (static_cast<size_t>(limb) << (LOGN + PRE));
    W *g = data + (vec << (LOGN + PRE)) + (static_cast<size_t>(sub) << LOGN);
    ntt_fwd_lazy_body<W, LOGN, LOGR, PRE, TIGHT, NT>(g, LoadData<W, NT>{g}, tw, lc, sub);
}

// digit d (shift, mask) of a coefficient-domain source vector: the gadget decomposition inside a transform's load
template <typename W, bool REDUCE>
struct LoadDigitOf {
    const W *src;
    uint32_t shift;
    W mask, q;
    __device__ __forceinline__ W operator()(uint32_t e) const {
        const W digit = (src[e] >> shift) & mask;
        if constexpr (REDUCE) return digit >= q ? digit % q : digit;
        else return digit;
    }
};
template <typename W, bool REDUCE>
__device__ __forceinline__ LoadDigitOf<W, REDUCE> load_digit_of(const W *src, const LimbConst *limbs, uint32_t t, uint32_t d,
                                                                uint32_t base_bits, W q) {
    const uint32_t src_bits = limbs[t].kbits, shift = d * base_bits;
    W mask = 0;
    if (shift < src_bits && shift < 8 * sizeof(W)) {
        const uint32_t rem = src_bits - shift;
        const uint32_t db = base_bits < rem ? base_bits : rem;
        mask = db >= 8 * sizeof(W) ? static_cast<W>(~static_cast<W>(0)) : static_cast<W>((static_cast<W>(1) << db) - 1);
    }
    return LoadDigitOf<W, REDUCE>{src, shift < 8 * sizeof(W) ? shift : 0u, mask, q};
}

// decompose + forward transform for the rings whose vector fits LDS (the 2^14 grouped kernel and the head kernel of the
// larger rings have their own forms): output vector (orow, col, limb), orow = r k + t dpt + d, is the transform of digit d
// of the tower-t coefficient residues of source entry (r, col).  One pass over the k-times larger digit matrix (written with
// non-temporal stores, NTS, when it is at least 1 GiB) instead of three.  grid = (L * src_cols, k, source rows)
template <typename W, int LOGN, int LOGR, int WAVES_PER_EU, bool TIGHT, bool REDUCE, bool NTS>
__global__ void __launch_bounds__(1 << (LOGN - LOGR), WAVES_PER_EU)
    ntt_fwd_lazy_digits_kernel(W *__restrict__ out, const W *__restrict__ coeff, const TwPair<W> *__restrict__ tw_all,
                               const LimbConst *__restrict__ limbs, uint32_t L, uint32_t src_cols, uint32_t dpt,
                               uint32_t base_bits, uint32_t k) {
    const uint32_t limb = blockIdx.x % L, col = blockIdx.x / L;
    const uint32_t td = blockIdx.y, t = td / dpt, d = td - t * dpt;
    const size_t r = blockIdx.z;
    const LimbConst lc = limbs[limb];
    const TwPair<W> *tw = tw_all + (static_cast<size_t>(limb) << LOGN);
    const W *src = coeff + (((r * src_cols + col) * L + t) << LOGN);
    W *g = out + (((((r * k + td) * src_cols + col) * L) + limb) << LOGN);
    ntt_fwd_lazy_body<W, LOGN, LOGR, 0, TIGHT, NTS>(g, load_digit_of<W, REDUCE>(src, limbs, t, d, base_bits, static_cast<W>(lc.q)), tw,
                                                     lc, 0u);
}

// PRE > 0: the last PRE stages (and the N^-1 scaling) are left to ntt_inv_tail_kernel; outputs stay below 2q
// MULW: data <- INTT(in o w), w = the residues [limb][2^(LOGN + PRE)] of a resident EVAL-form ring element, multiplied in
// the load (Montgomery product; `limbs` = ctx->d_limbs_r here, or in the tail kernel when PRE > 0)
template <typename W, int LOGN, int LOGR, int WAVES_PER_EU, int PRE = 0, bool TIGHT = false, bool NT = false, bool MULW = false>
__global__ void __launch_bounds__(1 << (LOGN - LOGR), WAVES_PER_EU)
    ntt_inv_lazy_kernel(W *__restrict__ data, const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs,
                        uint32_t L, const W *in = nullptr, const W *__restrict__ mulw = nullptr) {
    typedef NttLdsCfg<W, LOGN, LOGR, true> Cfg;
    constexpr uint32_t N = Cfg::N, T = Cfg::T;
    constexpr int P = Cfg::P, CLAST = Cfg::CLAST, R = 1 << LOGR;
    static_assert(P == 3, "three passes");
    constexpr int CAP = TIGHT ? kTightCap : 31;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W *x = reinterpret_cast<W *>(smem);
    const uint32_t tid = threadIdx.x;
    const size_t vec = blockIdx.x >> PRE;
    const uint32_t sub = blockIdx.x & ((1u << PRE) - 1u);
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q);
    const W muw = static_cast<W>(sizeof(W) == 4 ? lc.mu32 : lc.mu64);
    const TwPair<W> *tw = tw_all + (static_cast<size_t>(limb) << (LOGN + PRE));
    W *g = data + (vec << (LOGN + PRE)) + (static_cast<size_t>(sub) << LOGN);

    // HBM -> LDS, 16 bytes per lane
    constexpr int VN = 16 / sizeof(W);
    if constexpr (MULW) {
        static_assert(sizeof(W) == 4, "the fused product is a 32-bit Montgomery product");
        const W *src = in + (g - data);
        const W *wsrc = mulw + (static_cast<size_t>(limb) << (LOGN + PRE)) + (static_cast<size_t>(sub) << LOGN);
        const W qninv = neg_inv_pow2<W>(q);
#pragma unroll
        for (uint32_t jj = 0; jj < N / VN / T; ++jj) {
            const uint32_t i = tid + jj * T;
            W d[VN], wv[VN];
            nt_load16<NT, W>(d, src + static_cast<size_t>(i) * VN);
            nt_load16<false, W>(wv, wsrc + static_cast<size_t>(i) * VN);
#pragma unroll
            for (int e = 0; e < VN; ++e) d[e] = csub<W>(mont_mul_lazy(d[e], wv[e], q, qninv), q);  // canonical, as a load would give
            nt_load16<false, W>(&x[lds_pad_c(i * VN)], d);
        }
    } else {
#pragma unroll
        for (uint32_t jj = 0; jj < N / VN / T; ++jj) {
            const uint32_t i = tid + jj * T;
            nt_load16<NT, W>(&x[lds_pad_c(i * VN)], g + static_cast<size_t>(i) * VN);
        }
    }
    __syncthreads();
    {   // contiguous pass: stages [2 LOGR, LOGN) in GS order
        constexpr int G = 1 << (LOGR - CLAST), E = 1 << CLAST;
        const uint32_t pb0 = lds_pad_c(tid * R);
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            W v[E];
#pragma unroll
            for (int u = 0; u < E; ++u) v[u] = x[pb0 + gi * E + u];
            gs_network_lazy<W, CLAST, false, CAP>(v, tw, (sub << (2 * LOGR)) + tid * G + gi, PRE + 2 * LOGR, q, lc);
            gs_fold<W, CLAST, CAP>(v, q, muw);
#pragma unroll
            for (int u = 0; u < E; ++u) x[pb0 + gi * E + u] = v[u];
        }
    }
    __syncthreads();
    {   // middle pass: stages [LOGR, 2 LOGR)
        constexpr uint32_t B = 1u << (LOGN - LOGR), S = B >> LOGR;
        uint32_t bi, pb;
        lds_set_addr<W, LOGN, LOGR, LOGR, LOGR>(tid, 0, bi, pb);
        W v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = x[pb + lds_pad_c(S * u)];
        gs_network_lazy<W, LOGR, false, CAP>(v, tw, (sub << LOGR) + bi, PRE + LOGR, q, lc);
        gs_fold<W, LOGR, CAP>(v, q, muw);
#pragma unroll
        for (int u = 0; u < R; ++u) x[pb + lds_pad_c(S * u)] = v[u];
    }
    __syncthreads();
    {   // strided pass: stages [0, LOGR), results straight to HBM (coalesced per u)
        const uint32_t pb = lds_pad_c(tid);
        W v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) v[u] = x[pb + lds_pad_c(T * u)];
        if constexpr (PRE == 0) {
            gs_network_lazy<W, LOGR, true, CAP>(v, tw, 0, 0, q, lc);
            // elements that did not go through the N^-1 Shoup product of the last stage carry
            // A-path bounds from the earlier stages of this pass; those that did are < 2q
#pragma unroll
            for (int u = 0; u < R; ++u) nt_store<NT, W>(csub<W>(v[u], q), g + tid + T * u);
        } else {
            gs_network_lazy<W, LOGR, false, CAP>(v, tw, sub, PRE, q, lc);
            gs_fold<W, LOGR, CAP>(v, q, muw);  // below 2q: the input bound of the tail kernel's butterflies
#pragma unroll
            for (int u = 0; u < R; ++u) nt_store<NT, W>(v[u], g + tid + T * u);
        }
    }
}


// ---- rings too large for LDS (2^16, 2^17 points; 64-bit words from 2^15): two kernels, two HBM round trips ----------
// Head: stages [0, PRE) on the strided sets {j + (N >> PRE) u}; afterwards the vector is 2^PRE independent sub-vectors,
// which ntt_fwd_lazy_kernel<.., PRE> transforms in LDS.  The inverse mirrors it: sub-vectors first (outputs below 2q),
// then the tail below with the N^-1 scaling.  (One launch per stage, 16-17 round trips, ran at 0.3-0.4 TB/s.)
template <typename W, int PRE>
__global__ void __launch_bounds__(256)
    ntt_fwd_head_kernel(W *__restrict__ data, const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs,
                        uint32_t L, uint32_t logN) {
    constexpr int R = 1 << PRE;
    const uint32_t S = (1u << logN) >> PRE;                   // stride between the elements of a set
    const uint32_t sets_blocks = S / blockDim.x;              // workgroups per vector
    const size_t vec = blockIdx.x / sets_blocks;
    const uint32_t j = (blockIdx.x - static_cast<uint32_t>(vec) * sets_blocks) * blockDim.x + threadIdx.x;
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q), twoq = q + q;
    const TwPair<W> *tw = tw_all + (static_cast<size_t>(limb) << logN);
    W *g = data + (vec << logN) + j;
    W v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = g[static_cast<size_t>(S) * u];
    ct_network_lazy<W, PRE>(v, tw, 0, 0, q, twoq);
#pragma unroll
    for (int u = 0; u < R; ++u) g[static_cast<size_t>(S) * u] = v[u];
}

// The head kernel with the gadget decomposition in its load (decompose.hip): output vector (orow, col, limb), orow =
// r k + t dpt + d, starts from digit d of the tower-t coefficient residues of source entry (r, col) instead of from its
// own contents.  Two-step decomposition at these sizes costs five passes over the k-times larger digit matrix (digits
// written, head read + write, sub-vectors read + write); this way three.
// grid = (sets_blocks * L * src_cols, k, source rows)
template <typename W, int PRE, bool REDUCE, bool NTS>
__global__ void __launch_bounds__(256)
    ntt_fwd_head_digits_kernel(W *__restrict__ out, const W *__restrict__ coeff, const TwPair<W> *__restrict__ tw_all,
                               const LimbConst *__restrict__ limbs, uint32_t L, uint32_t logN, uint32_t src_cols, uint32_t dpt,
                               uint32_t base_bits, uint32_t k) {
    constexpr int R = 1 << PRE;
    const uint32_t S = (1u << logN) >> PRE;
    const uint32_t sets_blocks = S / blockDim.x;
    const uint32_t sb = blockIdx.x % sets_blocks, rest = blockIdx.x / sets_blocks;
    const uint32_t limb = rest % L, col = rest / L;
    const uint32_t td = blockIdx.y, t = td / dpt, d = td - t * dpt;
    const size_t r = blockIdx.z;
    const uint32_t j = sb * blockDim.x + threadIdx.x;
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q), twoq = q + q;
    const TwPair<W> *tw = tw_all + (static_cast<size_t>(limb) << logN);
    const uint32_t src_bits = limbs[t].kbits, shift = d * base_bits;
    W mask = 0;
    if (shift < src_bits && shift < 8 * sizeof(W)) {
        const uint32_t rem = src_bits - shift;
        const uint32_t db = base_bits < rem ? base_bits : rem;
        mask = db >= 8 * sizeof(W) ? static_cast<W>(~static_cast<W>(0)) : static_cast<W>((static_cast<W>(1) << db) - 1);
    }
    const uint32_t sh = shift < 8 * sizeof(W) ? shift : 0;
    const W *src = coeff + (((r * src_cols + col) * L + t) << logN) + j;
    W *g = out + (((((r * k + td) * src_cols + col) * L) + limb) << logN) + j;
    W v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const W digit = (src[static_cast<size_t>(S) * u] >> sh) & mask;
        if constexpr (REDUCE) v[u] = digit >= q ? digit % q : digit;
        else v[u] = digit;
    }
    ct_network_lazy<W, PRE>(v, tw, 0, 0, q, twoq);
#pragma unroll
    for (int u = 0; u < R; ++u) nt_store<NTS, W>(v[u], g + static_cast<size_t>(S) * u);  // NTS (large outputs): keeps the source vectors (re-read by L * dpt workgroups) in L2
}

template <typename W, int PRE, bool TIGHT = false>
__global__ void __launch_bounds__(256)
    ntt_inv_tail_kernel(W *__restrict__ data, const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs,
                        uint32_t L, uint32_t logN) {
    constexpr int R = 1 << PRE;
    const uint32_t S = (1u << logN) >> PRE;
    const uint32_t sets_blocks = S / blockDim.x;
    const size_t vec = blockIdx.x / sets_blocks;
    const uint32_t j = (blockIdx.x - static_cast<uint32_t>(vec) * sets_blocks) * blockDim.x + threadIdx.x;
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q);
    const TwPair<W> *tw = tw_all + (static_cast<size_t>(limb) << logN);
    W *g = data + (vec << logN) + j;
    W v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) v[u] = g[static_cast<size_t>(S) * u];
    gs_network_lazy<W, PRE, true, TIGHT ? kTightCap : 31>(v, tw, 0, 0, q, lc);
#pragma unroll
    for (int u = 0; u < R; ++u) g[static_cast<size_t>(S) * u] = csub<W>(v[u], q);
}
