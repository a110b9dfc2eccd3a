// ntt_lds_u64.hip — instantiations of the lazy LDS NTT for 64-bit residue words.
#include "ntt14.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

typedef uint64_t W;
#include "ntt_lds_dispatch.inc"
#include "ntt_f64.h"

// ---- moduli below 2^51: the double-precision transforms (ntt_f64.h), whole-vector sizes 2^10..2^14 ---------------------
static bool f64_path(const GpuContext *ctx) {
    return ctx->f64_ok && !ctx->env.ntt64_int && ctx->env.ntt_path <= 1 && ctx->logN >= 10 && ctx->logN <= 17;
}

template <int LOGN, int LOGR, int WPE, int ELIM>
static int launch_f64(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse) {
    const size_t lds = lds_padded_words(size_t(1) << LOGN) * sizeof(double);
    const bool nt = (vectors << LOGN) * sizeof(uint64_t) >= (size_t(1) << 30);  // as launch_lazy: batches no cache holds
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (lds > 64 * 1024 && !(configured.load() & bit)) {
        const void *fns[] = {reinterpret_cast<const void *>(nttf::fwd_kernel<LOGN, LOGR, WPE, ELIM, false>),
                             reinterpret_cast<const void *>(nttf::fwd_kernel<LOGN, LOGR, WPE, ELIM, true>),
                             reinterpret_cast<const void *>(nttf::inv_kernel<LOGN, LOGR, WPE, ELIM, false>),
                             reinterpret_cast<const void *>(nttf::inv_kernel<LOGN, LOGR, WPE, ELIM, true>)};
        for (const void *f : fns) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        configured.fetch_or(bit);
    }
    const dim3 grid(static_cast<unsigned>(vectors)), block(1u << (LOGN - LOGR));
    const TwF *tw = static_cast<const TwF *>(inverse ? ctx->d_twf_inv : ctx->d_twf_fwd);
    const F64Limb *fl = static_cast<const F64Limb *>(ctx->d_flimbs);
    if (!inverse) {
        if (nt) MXX_LAUNCH((nttf::fwd_kernel<LOGN, LOGR, WPE, ELIM, true>), grid, block, lds, ctx->stream, data, tw, fl, L);
        else MXX_LAUNCH((nttf::fwd_kernel<LOGN, LOGR, WPE, ELIM, false>), grid, block, lds, ctx->stream, data, tw, fl, L);
    } else {
        if (nt) MXX_LAUNCH((nttf::inv_kernel<LOGN, LOGR, WPE, ELIM, true>), grid, block, lds, ctx->stream, data, tw, fl, L);
        else MXX_LAUNCH((nttf::inv_kernel<LOGN, LOGR, WPE, ELIM, false>), grid, block, lds, ctx->stream, data, tw, fl, L);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// 2^(SUBLOG + PRE) points: head / tail kernel on the strided sets + the LDS kernels on the 2^PRE sub-vectors (folded doubles
// travel between the two launches in the vector's own 8-byte slots)
template <int SUBLOG, int LOGR, int WPE, int PRE, int ELIM>
static int launch_f64_split(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse) {
    const size_t lds = lds_padded_words(size_t(1) << SUBLOG) * sizeof(double);
    if (vectors > (0x7fffffffull >> PRE)) return -1;
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (lds > 64 * 1024 && !(configured.load() & bit)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(nttf::fwd_kernel<SUBLOG, LOGR, WPE, ELIM, false, PRE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(nttf::inv_kernel<SUBLOG, LOGR, WPE, ELIM, false, PRE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        configured.fetch_or(bit);
    }
    const uint32_t logN = SUBLOG + PRE;
    const dim3 sub_grid(static_cast<unsigned>(vectors << PRE)), sub_block(1u << (SUBLOG - LOGR));
    const dim3 set_grid(static_cast<unsigned>(vectors * (((1u << logN) >> PRE) / 256u))), set_block(256);
    const F64Limb *fl = static_cast<const F64Limb *>(ctx->d_flimbs);
    if (!inverse) {
        const TwF *tw = static_cast<const TwF *>(ctx->d_twf_fwd);
        MXX_LAUNCH((nttf::head_kernel<PRE, ELIM, false, false, false>), set_grid, set_block, 0, ctx->stream, data, data, tw, fl, ctx->d_limbs,
                   L, logN, 0u, 0u, 0u, 0u);
        MXX_LAUNCH((nttf::fwd_kernel<SUBLOG, LOGR, WPE, ELIM, false, PRE>), sub_grid, sub_block, lds, ctx->stream, data, tw, fl, L);
    } else {
        const TwF *tw = static_cast<const TwF *>(ctx->d_twf_inv);
        MXX_LAUNCH((nttf::inv_kernel<SUBLOG, LOGR, WPE, ELIM, false, PRE>), sub_grid, sub_block, lds, ctx->stream, data, tw, fl, L);
        MXX_LAUNCH((nttf::tail_kernel<PRE, ELIM>), set_grid, set_block, 0, ctx->stream, data, tw, fl, L, logN);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ELIM: the largest bound (units of q / 4) a value may reach: |x| < 2^53 means < 4 q at 51 bits, < 16 q below 2^49, and
// below 2^40 no stage of a pass ever needs a fold
template <int ELIM>
static int dispatch_f64(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse) {
    switch (ctx->logN) {
        case 15: return launch_f64_split<11, 4, 1, 4, ELIM>(ctx, data, vectors, L, inverse);
        case 16: return launch_f64_split<12, 4, 1, 4, ELIM>(ctx, data, vectors, L, inverse);
        case 17: return launch_f64_split<12, 4, 1, 5, ELIM>(ctx, data, vectors, L, inverse);
        case 10: return launch_f64<10, 4, 1, ELIM>(ctx, data, vectors, L, inverse);
        case 11: return launch_f64<11, 4, 1, ELIM>(ctx, data, vectors, L, inverse);
        case 12: return launch_f64<12, 4, 1, ELIM>(ctx, data, vectors, L, inverse);
        case 13: return launch_f64<13, 5, 1, ELIM>(ctx, data, vectors, L, inverse);
        case 14: return launch_f64<14, 5, 2, ELIM>(ctx, data, vectors, L, inverse);
        default: return -1;
    }
}

// rings below 2^10 points with moduli below 2^51: the double-precision form of the generic one-stage-per-barrier kernel
static int launch_f64_small(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse) {
    const uint32_t logN = ctx->logN;
    const size_t N = size_t(1) << logN;
    unsigned threads = static_cast<unsigned>(N / 2);
    if (threads < 64) threads = 64;
    if (threads > 512) threads = 512;
    const dim3 grid(static_cast<unsigned>(vectors)), block(threads);
    const size_t lds = N * sizeof(double);
    const F64Limb *fl = static_cast<const F64Limb *>(ctx->d_flimbs);
    if (!inverse)
        MXX_LAUNCH((nttf::small_kernel<false>), grid, block, lds, ctx->stream, data, static_cast<const TwF *>(ctx->d_twf_fwd), fl, L, logN);
    else
        MXX_LAUNCH((nttf::small_kernel<true>), grid, block, lds, ctx->stream, data, static_cast<const TwF *>(ctx->d_twf_inv), fl, L, logN);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_ntt_lds_u64(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse) {
    if (ctx->f64_ok && !ctx->env.ntt64_int && ctx->env.ntt_path <= 1 && ctx->logN >= 1 && ctx->logN < 10 && vectors <= 0x7fffffffull)
        return launch_f64_small(ctx, data, vectors, L, inverse);
    if (f64_path(ctx) && vectors <= 0x7fffffffull) {
        const int rc = ctx->crt_bits <= 40   ? dispatch_f64<4095>(ctx, data, vectors, L, inverse)
                       : ctx->crt_bits <= 49 ? dispatch_f64<63>(ctx, data, vectors, L, inverse)
                                             : dispatch_f64<15>(ctx, data, vectors, L, inverse);
        if (rc >= 0) return rc;
    }
    return dispatch_ntt_lds(ctx, data, vectors, L, inverse);
}

template <int LOGN, int LOGR, int WPE, int ELIM>
static int launch_f64_digits(GpuContext *ctx, uint64_t *out, const uint64_t *coeff, uint32_t L, uint32_t src_cols, size_t src_rows,
                             uint32_t dpt, uint32_t base_bits, size_t k, bool reduce) {
    const size_t lds = lds_padded_words(size_t(1) << LOGN) * sizeof(double);
    const size_t vectors = src_rows * k * src_cols * L;
    const bool nts = (vectors << LOGN) * sizeof(uint64_t) >= (size_t(1) << 29);
    const void *fs[4] = {reinterpret_cast<const void *>(nttf::fwd_digits_kernel<LOGN, LOGR, WPE, ELIM, false, false>),
                         reinterpret_cast<const void *>(nttf::fwd_digits_kernel<LOGN, LOGR, WPE, ELIM, false, true>),
                         reinterpret_cast<const void *>(nttf::fwd_digits_kernel<LOGN, LOGR, WPE, ELIM, true, false>),
                         reinterpret_cast<const void *>(nttf::fwd_digits_kernel<LOGN, LOGR, WPE, ELIM, true, true>)};
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (lds > 64 * 1024 && !(configured.load() & bit)) {
        for (const void *f : fs) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        configured.fetch_or(bit);
    }
    const uint64_t gx = static_cast<uint64_t>(L) * src_cols;
    if (gx > 0x7fffffffull || k > 65535 || src_rows > 65535) return -1;
    const dim3 grid(static_cast<unsigned>(gx), static_cast<unsigned>(k), static_cast<unsigned>(src_rows)), block(1u << (LOGN - LOGR));
    const TwF *tw = static_cast<const TwF *>(ctx->d_twf_fwd);
    const F64Limb *fl = static_cast<const F64Limb *>(ctx->d_flimbs);
#define MXX_F64D(RED, NTSF)                                                                                                    \
    MXX_LAUNCH((nttf::fwd_digits_kernel<LOGN, LOGR, WPE, ELIM, RED, NTSF>), grid, block, lds, ctx->stream, out, coeff, tw, fl,  \
               ctx->d_limbs, L, src_cols, dpt, base_bits, static_cast<uint32_t>(k))
    if (reduce) {
        if (nts) MXX_F64D(true, true);
        else MXX_F64D(true, false);
    } else {
        if (nts) MXX_F64D(false, true);
        else MXX_F64D(false, false);
    }
#undef MXX_F64D
    HIP_TRY(hipGetLastError());
    return 0;
}

// decompose + forward transform at the split sizes: head kernel with the digits in its load, then the sub-vectors
template <int SUBLOG, int LOGR, int WPE, int PRE, int ELIM>
static int launch_f64_split_digits(GpuContext *ctx, uint64_t *out, const uint64_t *coeff, uint32_t L, uint32_t src_cols, size_t src_rows,
                                   uint32_t dpt, uint32_t base_bits, size_t k, bool reduce) {
    const size_t lds = lds_padded_words(size_t(1) << SUBLOG) * sizeof(double);
    const size_t vectors = src_rows * k * src_cols * L;
    if (vectors > (0x7fffffffull >> PRE)) return -1;
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (lds > 64 * 1024 && !(configured.load() & bit)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(nttf::fwd_kernel<SUBLOG, LOGR, WPE, ELIM, false, PRE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        configured.fetch_or(bit);
    }
    const uint32_t logN = SUBLOG + PRE;
    const uint64_t gx = static_cast<uint64_t>(((1u << logN) >> PRE) / 256u) * L * src_cols;
    if (gx > 0x7fffffffull || k > 65535 || src_rows > 65535) return -1;
    const dim3 set_grid(static_cast<unsigned>(gx), static_cast<unsigned>(k), static_cast<unsigned>(src_rows));
    const dim3 sub_grid(static_cast<unsigned>(vectors << PRE)), sub_block(1u << (SUBLOG - LOGR));
    const bool nts = (vectors << logN) * sizeof(uint64_t) >= (size_t(1) << 29);
    const TwF *tw = static_cast<const TwF *>(ctx->d_twf_fwd);
    const F64Limb *fl = static_cast<const F64Limb *>(ctx->d_flimbs);
#define MXX_F64H(RED, NTSF)                                                                                                     \
    MXX_LAUNCH((nttf::head_kernel<PRE, ELIM, true, RED, NTSF>), set_grid, dim3(256), 0, ctx->stream, out, coeff, tw, fl, ctx->d_limbs, \
               L, logN, src_cols, dpt, base_bits, static_cast<uint32_t>(k))
    if (reduce) {
        if (nts) MXX_F64H(true, true);
        else MXX_F64H(true, false);
    } else {
        if (nts) MXX_F64H(false, true);
        else MXX_F64H(false, false);
    }
#undef MXX_F64H
    MXX_LAUNCH((nttf::fwd_kernel<SUBLOG, LOGR, WPE, ELIM, false, PRE>), sub_grid, sub_block, lds, ctx->stream, out, tw, fl, L);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int ELIM>
static int dispatch_f64_digits(GpuContext *ctx, uint64_t *out, const uint64_t *coeff, uint32_t L, uint32_t src_cols, size_t src_rows,
                               uint32_t dpt, uint32_t base_bits, size_t k, bool reduce) {
#define MXX_ARGS ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce
    switch (ctx->logN) {
        case 15: return launch_f64_split_digits<11, 4, 1, 4, ELIM>(MXX_ARGS);
        case 16: return launch_f64_split_digits<12, 4, 1, 4, ELIM>(MXX_ARGS);
        case 17: return launch_f64_split_digits<12, 4, 1, 5, ELIM>(MXX_ARGS);
        case 10: return launch_f64_digits<10, 4, 1, ELIM>(MXX_ARGS);
        case 11: return launch_f64_digits<11, 4, 1, ELIM>(MXX_ARGS);
        case 12: return launch_f64_digits<12, 4, 1, ELIM>(MXX_ARGS);
        case 13: return launch_f64_digits<13, 5, 1, ELIM>(MXX_ARGS);
        case 14: return launch_f64_digits<14, 5, 2, ELIM>(MXX_ARGS);
        default: return -1;
    }
#undef MXX_ARGS
}

// decompose + forward transform in one pass for 64-bit words (see launch_ntt_digits_u32); -1: not available
int launch_ntt_digits_u64(GpuContext *ctx, uint64_t *out, const uint64_t *coeff, size_t out_vectors, uint32_t L,
                          uint32_t src_cols, uint32_t towers, uint32_t dpt, uint32_t base_bits, size_t k) {
    const EnvSwitches &env = ctx->env;
    if (!ctx->lazy_ok || env.ntt_path > 1 || !env.decompose_fused || out_vectors > 0x7fffffffull || k >> 32) return -1;
    if (k == 0 || src_cols == 0 || out_vectors % (k * src_cols * L) != 0) return -1;
    const size_t src_rows = out_vectors / (k * src_cols * L);
    if (src_rows > 65535 || k > 65535 || static_cast<uint64_t>(src_cols) * L > 0x7fffffffull) return -1;
    const uint32_t digit_bits = std::min<uint32_t>(base_bits, ctx->crt_bits);
    uint64_t min_q = ~0ull;
    for (uint32_t l = 0; l < L; ++l) min_q = std::min<uint64_t>(min_q, ctx->moduli[l]);
    const bool reduce = digit_bits >= 63 || ((1ull << digit_bits) - 1) >= min_q;
    (void)towers;
    if (f64_path(ctx)) {
        const int rc = ctx->crt_bits <= 40   ? dispatch_f64_digits<4095>(ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce)
                       : ctx->crt_bits <= 49 ? dispatch_f64_digits<63>(ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce)
                                             : dispatch_f64_digits<15>(ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce);
        if (rc >= 0) return rc;
    }
    return dispatch_ntt_digits(ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce);
}
