// ntt14.h — 2^14-point negacyclic NTT / INTT, "grouped" layout: 4 workgroups per CU.
//
// Why: with the whole 64 KB vector in LDS (ntt_lds.h) only two workgroups fit a CU; each one
// alternates a memory phase (64 KB in, 64 KB out at the CU's HBM share) with a VALU phase of
// about the same length, and two such customers keep two balanced servers ~2/3 busy
// (measured: 148 us with 80 us of memory skeleton and ~93 us of VALU).  Four customers give
// ~4/5.  The LDS footprint is cut to 18 KB per buffer by finishing stages 0..4 in registers
// first: after them the vector is 32 independent 512-point sub-transforms ("blocks"), and
// thread t holds element t of every block.  Blocks are then processed 8 at a time (one per
// wave): one transposing LDS write + ONE workgroup barrier per group; the nine remaining
// stages of a block (3 passes x 3 radix-2 stages) touch only that wave's 2 KB, so they need
// no workgroup barrier at all (LDS is in-order within a wave), and the finished block goes
// straight to HBM (a wave stores 2 KB contiguous).  The inverse mirrors it.
//
// Same lazy butterflies, twiddle tables and results as ntt_lds.h (bit-identical outputs).
#pragma once

#include "ntt_lds.h"

namespace ntt14 {
constexpr int LOGN = 14;
constexpr uint32_t N = 1u << LOGN;
constexpr uint32_t T = 512;                 // threads = 8 waves
constexpr int R0 = 32;                      // elements per thread in the register pass (5 stages)
constexpr uint32_t BLK = 512;               // sub-transform size after 5 stages
constexpr uint32_t BLK_PAD = BLK + 64;      // +4 words per 32 (see pad64)
constexpr uint32_t GROUP_WORDS = 8 * BLK_PAD;

// LDS position of element `pos` of a 512-point block: +4 words per 32.  The three access shapes of a block pass
// are "lane + 64 m", "64 c + j + 8 m" (both one dword per lane) and "8 contiguous words per lane" (which the
// compiler issues as two 128-bit accesses, with their own lane grouping: MI355X_MICROARCH.md, LDS table).  With the
// round-1 padding (+8 per 64) the 128-bit reads ran at 3x and the 128-bit writes at 2x their conflict-free
// cycles (SQ_LDS_BANK_CONFLICT = 15 % of SQ_LDS_IDX_ACTIVE); a bank simulation over paddings of the form
// "a words per 2^b" (tools/lds_pad_search.py) finds +4 per 32 the best: dword shapes and 128-bit writes
// conflict-free, 128-bit reads at 2x (an 8-word lane stride puts the 16 lanes of a 128-bit read group on even
// 4-bank slots only; no padding in that family separates them without breaking the dword shapes).
__device__ __forceinline__ uint32_t pad64(uint32_t pos) { return pos + ((pos >> 5) << 2); }

// LDS ops of one wave execute in order; this only stops the compiler from moving them
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// forward transform of one vector: `load(e)` supplies coefficient e (natural order), the result
// (canonical residues, bit-reversed order) goes to g
// TIGHT: 26..28-bit moduli (ntt_lds.h): the block passes first bring their inputs (below 16 q) under 8 q
// NTS: non-temporal stores of the finished blocks.  Right for LARGE digit transforms (the host sets it from 1 GiB of
// output: a k-times larger matrix that nothing reads back from cache, source vectors re-read by L * dpt workgroups: decompose 18.5 -> 17.9 ms, same-box A/B); wrong for
// the plain transform, whose output the next kernel often finds in the Infinity Cache (M1: the fused inverse behind a
// forward transform with such stores ran 158 -> 185 us).
// ADD: the vector `addv` (canonical residues, evaluation order) is added to the result on its way out
// GPUPOLY_PHASE_TIMING builds only (tools/build_variant.sh phase PHASE_TIMING=1; tools/ab_ntt_phases.sh; the mask comes from
// MXX_HIP_NTT_PHASE): `phase` switches parts of the kernel off at run time - bit 0: no global loads (synthetic inputs),
// bit 1: no butterflies, bit 2: no global stores, bit 5: no LDS traffic and no barriers, bit 6: twiddles from registers
// instead of the tables - to time the memory skeleton, the arithmetic and the exchanges separately.  Results are wrong by design in those modes.
#ifdef GPUPOLY_PHASE_TIMING
#define NTT14_PHASE(bit) ((phase & (bit)) != 0)
#else
#define NTT14_PHASE(bit) false
#endif
// COAL: the finished block takes one more trip through its LDS rows so that every store instruction of a wave covers 1 KB
// of CONTIGUOUS memory (lane l: 16 bytes at 16 l).  Without it a lane holds 8 consecutive words and its two 16-byte
// stores sit 32 bytes apart: each instruction half-fills 64 lines, and the two halves of a line reach the memory side as
// separate partial writes when the store is non-temporal (streams of 1 GiB and more, the digit transforms of a
// decomposition): "stores only" ran at 1.7 TB/s there (tools/ab_ntt_phases.sh).
template <typename W, bool TIGHT, bool NTS, typename Load, bool ADD = false, bool COAL = false, bool NOTW = false>
__device__ __forceinline__ void fwd_body(W *g, const Load load, const TwPair<W> *__restrict__ tw_all,
                                         const LimbConst &lc, uint32_t limb, const W *__restrict__ addv = nullptr,
                                         uint32_t phase = 0) {
    (void)phase;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W *xs = reinterpret_cast<W *>(smem);  // [2][8][BLK_PAD]
    constexpr int VN = 16 / sizeof(W);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const W q = static_cast<W>(lc.q), twoq = q + q;
    const W muw = static_cast<W>(sizeof(W) == 4 ? lc.mu32 : lc.mu64);
    const TwPair<W> *tw = tw_all + static_cast<size_t>(limb) * N;

    // stages 0..4 in registers: element tid of each of the 32 blocks
    W h[R0];
#pragma unroll
    for (int u = 0; u < R0; ++u) h[u] = NTT14_PHASE(1) ? static_cast<W>((tid * 2654435761u + u * 40503u + limb) & 0xffffu) : load(tid + T * u);
    if (!NTT14_PHASE(2)) ct_network_lazy<W, 5, NOTW>(h, tw, 0, 0, q, twoq);

#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        W *x = xs + (grp & 1) * GROUP_WORDS;
        // transpose through LDS: block (8*grp + m) receives its element `tid`
        if (!NTT14_PHASE(32)) {
#pragma unroll
            for (int m = 0; m < 8; ++m) x[m * BLK_PAD + pad64(tid)] = h[8 * grp + m];
            __syncthreads();
        }
        W *xb = x + wave * BLK_PAD;            // from here on this wave owns block B alone
        const uint32_t B = 8u * grp + wave;
        W v[8];
        {   // stages 5,6,7: sets {lane + 64 m}; twiddles are wave-uniform
            const uint32_t base = pad64(lane);  // pad64(lane + 64 m) = pad64(lane) + 72 m
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = NTT14_PHASE(32) ? h[8 * grp + m] : xb[base + 72 * m];  // bit 5: no LDS traffic, no barriers (arithmetic only)
            if (!NTT14_PHASE(2)) {
                ct_prefold<W, 3, TIGHT>(v, q);
                ct_network_lazy<W, 3, NOTW>(v, tw, B, 5, q, twoq);
            }
            if (!NTT14_PHASE(32)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) xb[base + 72 * m] = v[m];
            }
        }
        if (!NTT14_PHASE(32)) wave_sync();
        {   // stages 8,9,10: sets {64 c + j + 8 m}, c = lane/8, j = lane%8
            const uint32_t c = lane >> 3, j = lane & 7u;
            const uint32_t base = 72 * c + j;  // pad64(64 c + j + 8 m) = 72 c + j + 8 m + 4 (m >> 2)
            if (!NTT14_PHASE(32)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = xb[base + 8 * m + 4 * (m >> 2)];
            }
            if (!NTT14_PHASE(2)) {
                ct_prefold<W, 3, TIGHT>(v, q);
                ct_network_lazy<W, 3, NOTW>(v, tw, B * 8u + c, 8, q, twoq);
            }
            if (!NTT14_PHASE(32)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) xb[base + 8 * m + 4 * (m >> 2)] = v[m];
            }
        }
        if (!NTT14_PHASE(32)) wave_sync();
        {   // stages 11,12,13: 8 contiguous words per lane, then canonical form and out to HBM
            const uint32_t base = pad64(8 * lane);
            if (!NTT14_PHASE(32)) {
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = xb[base + m];
            }
            if (!NTT14_PHASE(2)) {
                ct_prefold<W, 3, TIGHT>(v, q);
                ct_network_lazy<W, 3, NOTW>(v, tw, B * 64u + lane, 11, q, twoq);
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = csub<W>(fold_2q<W>(v[m], q, muw), q);
            }
            if constexpr (COAL) {
                typedef W wx __attribute__((ext_vector_type(VN)));
                // back into the block's own rows (this wave's alone until the barrier two groups on), same slots they came from
#pragma unroll
                for (int m = 0; m < 8; m += VN) {
                    wx t;
#pragma unroll
                    for (int e = 0; e < VN; ++e) t[e] = v[m + e];
                    *reinterpret_cast<wx *>(&xb[base + m]) = t;
                }
                wave_sync();
                constexpr int CH = BLK / (64 * VN);  // store instructions per block: 2 for 32-bit words
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const uint32_t pos = c * 64 * VN + VN * lane;  // VN consecutive words never straddle a padding step (32 words)
                    wx t = *reinterpret_cast<const wx *>(&xb[pad64(pos)]);
                    W *dstc = g + B * BLK + pos;
                    if constexpr (ADD) {
                        const wx a = __builtin_nontemporal_load(reinterpret_cast<const wx *>(addv + B * BLK + pos));
#pragma unroll
                        for (int e = 0; e < VN; ++e) t[e] = csub<W>(t[e] + a[e], q);
                    }
                    if (NTT14_PHASE(4)) {
                        if (t[0] == static_cast<W>(0xdeadbeefu) && t[VN - 1] == static_cast<W>(0x12345u)) dstc[0] = t[0];
                    } else if constexpr (NTS) __builtin_nontemporal_store(t, reinterpret_cast<wx *>(dstc));
                    else *reinterpret_cast<wx *>(dstc) = t;
                }
                continue;
            }
            W *dst = g + B * BLK + 8 * lane;
#pragma unroll
            for (int m = 0; m < 8; m += VN) {
                typedef W wx __attribute__((ext_vector_type(VN)));
                wx t;
#pragma unroll
                for (int e = 0; e < VN; ++e) t[e] = v[m + e];
                if constexpr (ADD) {
                    const wx a = __builtin_nontemporal_load(reinterpret_cast<const wx *>(addv + B * BLK + 8 * lane + m));
#pragma unroll
                    for (int e = 0; e < VN; ++e) t[e] = csub<W>(t[e] + a[e], q);
                }
                if (NTT14_PHASE(4)) {
                    if (t[0] == static_cast<W>(0xdeadbeefu) && t[VN - 1] == static_cast<W>(0x12345u)) dst[m] = t[0];  // keeps the values alive
                } else if constexpr (NTS) __builtin_nontemporal_store(t, reinterpret_cast<wx *>(dst + m));
                else *reinterpret_cast<wx *>(dst + m) = t;
            }
        }
        // the other buffer is used next; this one is rewritten two groups later, behind the
        // next group's barrier
    }
}

template <typename W, bool NT = false>
struct LoadVector {
    const W *g;
    __device__ __forceinline__ W operator()(uint32_t e) const { return nt_load<NT, W>(g + e); }
};

// NT: non-temporal loads and stores of the data vector - for batches no cache level can hold (the launcher sets it from
// 1 GiB, as for the whole-vector LDS kernels): the stream then does not displace the twiddle tables
template <typename W, bool TIGHT = false, bool NT = false>
__global__ void __launch_bounds__(512, 8 / (sizeof(W) / 4))
    fwd_kernel(W *__restrict__ data, const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs,
               uint32_t L, uint32_t phase = 0) {
    // grid = (L, polys): the limb comes from the block index instead of a runtime division of it
    const uint32_t limb = blockIdx.x;
    const size_t vec = (static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * L + limb;
    const LimbConst lc = limbs[limb];
    W *g = data + vec * N;
#ifdef GPUPOLY_PHASE_TIMING
    // bit 3 of the phase mask selects the coalesced-store form, bit 4 cacheable stores, whatever the launcher chose
    if (phase & 64) {  // bit 6: twiddles from registers (no table loads)
        fwd_body<W, TIGHT, NT, LoadVector<W, NT>, false, NT, true>(g, LoadVector<W, NT>{g}, tw_all, lc, limb, nullptr, phase);
    } else if (phase & 8) {
        if (phase & 16) fwd_body<W, TIGHT, false, LoadVector<W, NT>, false, true>(g, LoadVector<W, NT>{g}, tw_all, lc, limb, nullptr, phase);
        else fwd_body<W, TIGHT, NT, LoadVector<W, NT>, false, true>(g, LoadVector<W, NT>{g}, tw_all, lc, limb, nullptr, phase);
    } else if (phase & 16) {
        fwd_body<W, TIGHT, false>(g, LoadVector<W, NT>{g}, tw_all, lc, limb, nullptr, phase);
    } else {
        fwd_body<W, TIGHT, NT>(g, LoadVector<W, NT>{g}, tw_all, lc, limb, nullptr, phase);
    }
#else
    (void)phase;
    // non-temporal streams (1 GiB and more) store through the coalescing LDS trip: see COAL
    fwd_body<W, TIGHT, NT, LoadVector<W, NT>, false, NT>(g, LoadVector<W, NT>{g}, tw_all, lc, limb);
#endif
}

// out = NTT(src) + add, out of place: the preimage's x = [.. ; p2 + z] takes the G-sampler's coefficient digits, the
// perturbation and the output block in ONE pass (gpupoly_matrix_ntt_add_rows) instead of a transform in place
// followed by an addition: 1.97 GB of traffic instead of 3.28 at M3A
template <typename W, bool TIGHT = false>
__global__ void __launch_bounds__(512, 8 / (sizeof(W) / 4))
    fwd_add_kernel(W *__restrict__ out, const W *__restrict__ src, const W *__restrict__ add,
                   const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs, uint32_t L) {
    const uint32_t limb = blockIdx.x;
    const size_t vec = (static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * L + limb;
    const LimbConst lc = limbs[limb];
    fwd_body<W, TIGHT, true, LoadVector<W, true>, true, true>(out + vec * N, LoadVector<W, true>{src + vec * N}, tw_all, lc, limb, add + vec * N);
}

// Gadget decomposition fused into the transform's load (decompose.hip): output vector
// (orow, col, limb) with orow = r*k + t*dpt + d is the NTT mod q_limb of the polynomial whose
// coefficients are digit d (shift, mask) of the tower-t residues of source entry (r, col).  The
// k-times larger digit matrix is written once, already in EVAL form, and never re-read.
// REDUCE: some digit can reach some output modulus (base wider than a limb): the loader then reduces; the common
// case (digits of base_bits <= bits of every limb) compiles to a shift and a mask per coefficient - the round-1
// kernel carried the modulo path and five runtime divisions of the block index in every thread, 1951 VALU
// instructions against the plain transform's 1406, 38.9 ns per vector against 27.3.
template <typename W, bool REDUCE>
struct LoadDigit {
    const W *src;  // coefficient-domain source vector (entry, tower)
    uint32_t shift;
    W mask, q;
    __device__ __forceinline__ W operator()(uint32_t e) const {
        const W digit = (src[e] >> shift) & mask;
        if constexpr (REDUCE) return digit >= q ? digit % q : digit;
        else return digit;
    }
};

// grid = (8 * L * ceil(src_cols / 8), k, source rows): blockIdx.x = (col % 8) + 8 * (limb + L * (col / 8)),
// blockIdx.y = t * dpt + d, blockIdx.z = r.  The L * dpt transforms that read one source vector (entry, tower t) then
// have the same block id modulo 8, i.e. run on one XCD and share its L2 (hardware places consecutive workgroup ids
// on consecutive XCDs; with limb fastest in x the eight limbs of a source landed on eight different L2s).
template <typename W, bool REDUCE, bool TIGHT = false, bool NTS = false>
__global__ void __launch_bounds__(512, 8 / (sizeof(W) / 4))
    fwd_digits_kernel(W *__restrict__ out, const W *__restrict__ coeff, const TwPair<W> *__restrict__ tw_all,
                      const LimbConst *__restrict__ limbs, uint32_t L, uint32_t src_cols, uint32_t towers, uint32_t dpt,
                      uint32_t base_bits, uint32_t k) {
    const uint32_t rest = blockIdx.x >> 3, col_hi = rest / L, limb = rest - col_hi * L;
    const uint32_t col = col_hi * 8u + (blockIdx.x & 7u);
    if (col >= src_cols) return;  // padding of the last group of 8 columns (whole workgroup)
    const uint32_t td = blockIdx.y;
    const size_t r = blockIdx.z;
    const uint32_t t = td / dpt, d = td - t * dpt;
    const size_t vec = ((r * k + td) * src_cols + col) * L + limb;
    const LimbConst lc = limbs[limb];
    const uint32_t src_bits = limbs[t].kbits, shift = d * base_bits;
    LoadDigit<W, REDUCE> load;
    load.src = coeff + ((r * src_cols + col) * L + t) * N;
    load.shift = shift < 8 * sizeof(W) ? shift : 0;
    load.mask = 0;
    if (shift < src_bits && shift < 8 * sizeof(W)) {
        const uint32_t rem = src_bits - shift;
        const uint32_t db = base_bits < rem ? base_bits : rem;
        load.mask = db >= 8 * sizeof(W) ? static_cast<W>(~static_cast<W>(0)) : static_cast<W>((static_cast<W>(1) << db) - 1);
    }
    load.q = static_cast<W>(lc.q);
    (void)towers;
    fwd_body<W, TIGHT, NTS, LoadDigit<W, REDUCE>, false, NTS>(out + vec * N, load, tw_all, lc, limb);
}

// SGN: the signed butterflies of ntt_lds.h (u32 words, q < 2^24, twiddle table ctx->d_tw2s_inv)
// MULW: the transform's load multiplies by a resident EVAL-form ring element first (mulw = its residues [limb][N],
// `in` = the EVAL operand): data <- INTT(in o w) with one kernel instead of a point-wise pass (a full HBM round trip)
// followed by the transform - gpupoly_matrix_mul_scalar_intt.  The product is a Montgomery one, REDC(x w) =
// x w 2^-32 (three multiply-class instructions, like a Shoup product, but on the plain 4-byte residue of w: round 2's
// Shoup pairs were 8 bytes per element - twice the data's own bytes, built by an extra launch - and their loads sat
// on the critical path of every group: 169 us against 120 for the plain inverse).  The missing factor 2^32 is folded
// into the last stage's constants: the MULW launch passes ctx->d_limbs_r, whose n_inv / inv_last_w are multiplied by
// 2^32 mod q.  PF: the next group's operands are requested before this group's butterflies start.
// CAP: bound-exponent cap of the unsigned butterflies (kTightCap for 26..28-bit moduli, ntt_lds.h)
// NT: non-temporal loads / stores of the data vector (batches of at least 1 GiB, set by the launcher)
template <typename W, bool SGN, bool MULW = false, int CAP = 31, int WPS = ((SGN && !MULW) ? 8 : 6), bool NT = false>
__global__ void __launch_bounds__(512, WPS / (sizeof(W) / 4))
    inv_kernel(W *__restrict__ data, const TwPair<W> *__restrict__ tw_all, const LimbConst *__restrict__ limbs,
               uint32_t L, const W *in = nullptr, const W *__restrict__ mulw = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W *xs = reinterpret_cast<W *>(smem);
    typedef typename std::conditional<sizeof(W) == 4, uint4, ulonglong2>::type V16;
    constexpr int VN = 16 / sizeof(W);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t limb = blockIdx.x;  // grid = (L, polys lo, polys hi)
    const size_t vec = (static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y) * L + limb;
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q);
    const W muw = static_cast<W>(sizeof(W) == 4 ? lc.mu32 : lc.mu64);
    const TwPair<W> *tw = tw_all + static_cast<size_t>(limb) * N;
    W *g = data + vec * N;
    const W *gin = (!MULW && in) ? in + vec * N : g;  // out of place without MULW: `in` must not overlap `data`

    // MULW: operands of the group after the current one, requested a group ahead
    V16 dnext[8 / VN], wnext[8 / VN];
    const W *src0 = (MULW ? in + vec * N : g) + wave * BLK + 8 * lane;
    const W *wsrc0 = MULW ? mulw + static_cast<size_t>(limb) * N + wave * BLK + 8 * lane : nullptr;
    if constexpr (MULW) {
#pragma unroll
        for (int m = 0; m < 8 / VN; ++m) {
            dnext[m] = *reinterpret_cast<const V16 *>(src0 + m * VN);
            wnext[m] = *reinterpret_cast<const V16 *>(wsrc0 + m * VN);
        }
    }

    W h[R0];
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        W *x = xs + (grp & 1) * GROUP_WORDS;
        W *xb = x + wave * BLK_PAD;
        const uint32_t B = 8u * grp + wave;
        W v[8];
        {   // stages 13,12,11 on 8 contiguous words per lane, straight from HBM
            if constexpr (MULW) {
                static_assert(sizeof(W) == 4, "the fused product is a 32-bit Montgomery product");
                W wv[8];
#pragma unroll
                for (int m = 0; m < 8 / VN; ++m) {
                    *reinterpret_cast<V16 *>(&v[m * VN]) = dnext[m];
                    *reinterpret_cast<V16 *>(&wv[m * VN]) = wnext[m];
                }
                if (grp < 3) {
#pragma unroll
                    for (int m = 0; m < 8 / VN; ++m) {
                        dnext[m] = *reinterpret_cast<const V16 *>(src0 + (grp + 1) * 8 * BLK + m * VN);
                        wnext[m] = *reinterpret_cast<const V16 *>(wsrc0 + (grp + 1) * 8 * BLK + m * VN);
                    }
                }
                const W qninv = neg_inv_pow2<W>(q);
#pragma unroll
                for (int m = 0; m < 8; ++m) v[m] = csub<W>(mont_mul_lazy(v[m], wv[m], q, qninv), q);  // canonical, as a load would give
            } else {
                const W *src = gin + B * BLK + 8 * lane;
#pragma unroll
                for (int m = 0; m < 8; m += VN) nt_load16<NT, W>(&v[m], src + m);
            }
            if constexpr (SGN) {  // canonical inputs (exponent 0); keep exponents <= 2
                gs_network_signed<3, false>(v, tw, B * 64u + lane, 11, q, lc);
                gs_fold_signed<3, 0, 2>(v, q, muw);
            } else {
                gs_network_lazy<W, 3, false, CAP>(v, tw, B * 64u + lane, 11, q, lc);
                gs_fold<W, 3, CAP>(v, q, muw);
            }
            const uint32_t base = pad64(8 * lane);
#pragma unroll
            for (int m = 0; m < 8; ++m) xb[base + m] = v[m];
        }
        wave_sync();
        {   // stages 10,9,8
            const uint32_t c = lane >> 3, j = lane & 7u;
            const uint32_t base = 72 * c + j;
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = xb[base + 8 * m + 4 * (m >> 2)];
            if constexpr (SGN) {  // inputs <= 2 (which of them depends on the lane): keep <= 3
                gs_network_signed<3, false>(v, tw, B * 8u + c, 8, q, lc);
                gs_fold_signed<3, 2, 3>(v, q, muw);
            } else {
                gs_network_lazy<W, 3, false, CAP>(v, tw, B * 8u + c, 8, q, lc);
                gs_fold<W, 3, CAP>(v, q, muw);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) xb[base + 8 * m + 4 * (m >> 2)] = v[m];
        }
        wave_sync();
        {   // stages 7,6,5 (wave-uniform twiddles)
            const uint32_t base = pad64(lane);
#pragma unroll
            for (int m = 0; m < 8; ++m) v[m] = xb[base + 72 * m];
            if constexpr (SGN) {  // inputs <= 3 (butterfly inputs reach 5); the register pass wants 1
                gs_network_signed<3, false>(v, tw, B, 5, q, lc);
                gs_fold_signed<3, 3, 1>(v, q, muw);
            } else {
                gs_network_lazy<W, 3, false, CAP>(v, tw, B, 5, q, lc);
                gs_fold<W, 3, CAP>(v, q, muw);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) xb[base + 72 * m] = v[m];
        }
        __syncthreads();
        // transpose back: element `tid` of each of the group's 8 blocks
#pragma unroll
        for (int m = 0; m < 8; ++m) h[8 * grp + m] = x[m * BLK_PAD + pad64(tid)];
    }
    // stages 4..0 in registers (N^-1 folded into the last one), coalesced per u
    if constexpr (SGN) gs_network_signed<5, true>(h, tw, 0, 0, q, lc);
    else gs_network_lazy<W, 5, true, CAP>(h, tw, 0, 0, q, lc);
#pragma unroll
    for (int u = 0; u < R0; ++u) nt_store<NT, W>(csub<W>(h[u], q), g + tid + T * u);
}

static inline size_t lds_bytes(size_t word) { return 2 * GROUP_WORDS * word; }
}  // namespace ntt14
