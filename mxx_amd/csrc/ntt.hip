// ntt.hip — negacyclic NTT / INTT over Z_q[x]/(x^N+1), one workgroup per
// (polynomial, RNS limb) with the whole residue vector resident in LDS.
//
// Convention = the reference CPU path's (OpenFHE), restated in-tree by the
// reference at src/gadgets/ntt/mod.rs:189-198,288-392:
//   forward  : Cooley-Tukey, natural-order input -> bit-reversed evaluations,
//              stage m = 2^s uses fwd[m+i] for group i (fwd[bitrev(j)] = psi^j)
//   inverse  : Gentleman-Sande with inv[m+i], then * N^-1.
// It replaces gpu_matrix_ntt_all / gpu_matrix_intt_all of
// cuda/src/matrix/MatrixNTT.cu:787-811 (2 + log2 N launches and ~17 HBM round
// trips per transform there; one launch and one HBM round trip here).
//
// Three kernel families, all bit-identical to the CPU oracle:
//   * ntt_lds.h        : tuned lazy-butterfly LDS kernels, logN in [10,15] (hot path)
//   * generic (below)  : any logN that fits LDS, one radix-2 stage per barrier
//   * global (below)   : vectors larger than LDS, one stage per launch
#include "common.h"
#include "modarith.h"

#include <atomic>
#include <cstdlib>

// Any logN whose vector fits LDS: one radix-2 stage per barrier (small / odd sizes).
template <typename W, bool INV>
__global__ void ntt_generic_kernel(W *__restrict__ data, const W *__restrict__ tw_all, const W *__restrict__ tws_all,
                                   const LimbConst *__restrict__ limbs, uint32_t L, uint32_t logN) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    W *x = reinterpret_cast<W *>(smem);
    const uint32_t N = 1u << logN;
    const uint32_t tid = threadIdx.x, T = blockDim.x;
    const size_t vec = blockIdx.x;
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const W q = static_cast<W>(limbs[limb].q);
    const W *tw = tw_all + static_cast<size_t>(limb) * N;
    const W *tws = tws_all + static_cast<size_t>(limb) * N;
    W *g = data + vec * N;
    for (uint32_t i = tid; i < N; i += T) x[i] = g[i];
    __syncthreads();
    if (!INV) {
        uint32_t logt = logN - 1;
        for (uint32_t m = 1; m < N; m <<= 1, --logt) {
            const uint32_t t = 1u << logt;
            for (uint32_t b = tid; b < N / 2; b += T) {
                const uint32_t i = b >> logt, j = b & (t - 1);
                const uint32_t lo = (i << (logt + 1)) + j, hi = lo + t;
                const W U = x[lo];
                const W V = mul_shoup<W>(x[hi], tw[m + i], tws[m + i], q);
                x[lo] = add_mod<W>(U, V, q);
                x[hi] = sub_mod<W>(U, V, q);
            }
            __syncthreads();
        }
        for (uint32_t i = tid; i < N; i += T) g[i] = x[i];
    } else {
        uint32_t logt = 0;
        for (uint32_t m = N >> 1; m >= 1; m >>= 1, ++logt) {
            const uint32_t t = 1u << logt;
            for (uint32_t b = tid; b < N / 2; b += T) {
                const uint32_t i = b >> logt, j = b & (t - 1);
                const uint32_t lo = (i << (logt + 1)) + j, hi = lo + t;
                const W U = x[lo], V = x[hi];
                x[lo] = add_mod<W>(U, V, q);
                x[hi] = mul_shoup<W>(sub_mod<W>(U, V, q), tw[m + i], tws[m + i], q);
            }
            __syncthreads();
        }
        const W ninv = static_cast<W>(limbs[limb].n_inv);
        const W ninv_sh = static_cast<W>(limbs[limb].n_inv_sh);
        for (uint32_t i = tid; i < N; i += T) g[i] = mul_shoup<W>(x[i], ninv, ninv_sh, q);
    }
}

// Vectors too large for LDS: one radix-2 stage per launch straight on HBM.
template <typename W, bool INV>
__global__ void ntt_stage_global_kernel(W *__restrict__ data, const W *__restrict__ tw_all,
                                        const W *__restrict__ tws_all, const LimbConst *__restrict__ limbs, uint32_t L,
                                        uint32_t logN, uint32_t m, uint32_t logt, size_t vectors) {
    const size_t half = static_cast<size_t>(1) << (logN - 1);
    size_t idx = item_index();
    if (idx >= vectors * half) return;
    const size_t vec = idx >> (logN - 1);
    const uint32_t b = static_cast<uint32_t>(idx & (half - 1));
    const uint32_t limb = static_cast<uint32_t>(vec % L);
    const W q = static_cast<W>(limbs[limb].q);
    const size_t N = static_cast<size_t>(1) << logN;
    const W *tw = tw_all + limb * N;
    const W *tws = tws_all + limb * N;
    W *x = data + vec * N;
    const uint32_t t = 1u << logt;
    const uint32_t i = b >> logt, j = b & (t - 1);
    const uint32_t lo = (i << (logt + 1)) + j, hi = lo + t;
    const W U = x[lo];
    if (!INV) {
        const W V = mul_shoup<W>(x[hi], tw[m + i], tws[m + i], q);
        x[lo] = add_mod<W>(U, V, q);
        x[hi] = sub_mod<W>(U, V, q);
    } else {
        const W V = x[hi];
        x[lo] = add_mod<W>(U, V, q);
        x[hi] = mul_shoup<W>(sub_mod<W>(U, V, q), tw[m + i], tws[m + i], q);
    }
}

template <typename W>
__global__ void ntt_scale_global_kernel(W *__restrict__ data, const LimbConst *__restrict__ limbs, uint32_t L,
                                        uint32_t logN, size_t total) {
    size_t idx = item_index();
    if (idx >= total) return;
    const uint32_t limb = static_cast<uint32_t>((idx >> logN) % L);
    const W q = static_cast<W>(limbs[limb].q);
    data[idx] = mul_shoup<W>(data[idx], static_cast<W>(limbs[limb].n_inv), static_cast<W>(limbs[limb].n_inv_sh), q);
}

// ---- host-side dispatch -------------------------------------------------------------------
static constexpr size_t kMaxLdsBytes = 160 * 1024;

template <typename W, bool INV>
static int launch_generic(GpuContext *ctx, W *data, size_t vectors, uint32_t L) {
    const uint32_t logN = ctx->logN;
    const size_t N = size_t(1) << logN;
    const size_t lds = N * sizeof(W);
    auto kern = ntt_generic_kernel<W, INV>;
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
    }
    unsigned threads = static_cast<unsigned>(N / 2);
    if (threads < 64) threads = 64;
    if (threads > 512) threads = 512;
    MXX_LAUNCH((ntt_generic_kernel<W, INV>), dim3(static_cast<unsigned>(vectors)), dim3(threads), lds, ctx->stream, data,
                       static_cast<const W *>(INV ? ctx->d_tw_inv : ctx->d_tw_fwd),
                       static_cast<const W *>(INV ? ctx->d_tw_inv_sh : ctx->d_tw_fwd_sh), ctx->d_limbs, L, logN);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename W, bool INV>
static int launch_global(GpuContext *ctx, W *data, size_t vectors, uint32_t L) {
    const uint32_t logN = ctx->logN;
    const size_t half = size_t(1) << (logN - 1);
    const size_t total = vectors * half;
    const dim3 blocks = item_grid(total, 256);
    const W *tw = static_cast<const W *>(INV ? ctx->d_tw_inv : ctx->d_tw_fwd);
    const W *tws = static_cast<const W *>(INV ? ctx->d_tw_inv_sh : ctx->d_tw_fwd_sh);
    if (!INV) {
        uint32_t logt = logN - 1;
        for (uint32_t m = 1; m < (1u << logN); m <<= 1, --logt) {
            MXX_LAUNCH((ntt_stage_global_kernel<W, false>), blocks, dim3(256), 0, ctx->stream, data, tw,
                               tws, ctx->d_limbs, L, logN, m, logt, vectors);
        }
    } else {
        uint32_t logt = 0;
        for (uint32_t m = 1u << (logN - 1); m >= 1; m >>= 1, ++logt) {
            MXX_LAUNCH((ntt_stage_global_kernel<W, true>), blocks, dim3(256), 0, ctx->stream, data, tw,
                               tws, ctx->d_limbs, L, logN, m, logt, vectors);
        }
        const size_t words = vectors << logN;
        MXX_LAUNCH(ntt_scale_global_kernel<W>, item_grid(words, 256), dim3(256), 0,
                           ctx->stream, data, ctx->d_limbs, L, logN, words);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename W, bool INV>
static int launch_ntt_typed(GpuContext *ctx, W *data, size_t vectors, uint32_t L) {
    const uint32_t logN = ctx->logN;
    const size_t N = size_t(1) << logN;
    const int force = ctx->env.ntt_path;  // MXX_HIP_NTT_PATH = lds | generic | global (tests exercise every path)
    const bool fits_generic = N * sizeof(W) <= kMaxLdsBytes && logN >= 1;
    if (force == 3) return launch_global<W, INV>(ctx, data, vectors, L);
    if (force != 2 && (ctx->lazy_ok || ctx->tight_ok)) {  // tuned kernels: whole vector in LDS, or head / tail + sub-vectors beyond it
        int rc;
        if constexpr (sizeof(W) == 4) rc = launch_ntt_lds_u32(ctx, data, vectors, L, INV);
        else rc = launch_ntt_lds_u64(ctx, data, vectors, L, INV);
        if (rc >= 0) return rc;  // -1: no tuned kernel for this logN
    }
    if (!fits_generic) return launch_global<W, INV>(ctx, data, vectors, L);  // vector larger than LDS
    return launch_generic<W, INV>(ctx, data, vectors, L);
}

int launch_ntt(GpuContext *ctx, void *data, size_t vectors, int limbs_per_poly, bool inverse) {
    if (vectors == 0) return 0;
    if (vectors > 0x7fffffffull) return set_error("ntt: too many vectors for one launch");
    const uint32_t L = static_cast<uint32_t>(limbs_per_poly);
    MXX_TRACE_BYTES(2.0 * vectors * ctx->N * ctx->word_bytes);  // read once + write once (all launches of a split transform)
    if (ctx->wide) {
        return inverse ? launch_ntt_typed<uint64_t, true>(ctx, static_cast<uint64_t *>(data), vectors, L)
                       : launch_ntt_typed<uint64_t, false>(ctx, static_cast<uint64_t *>(data), vectors, L);
    }
    return inverse ? launch_ntt_typed<uint32_t, true>(ctx, static_cast<uint32_t *>(data), vectors, L)
                   : launch_ntt_typed<uint32_t, false>(ctx, static_cast<uint32_t *>(data), vectors, L);
}

// ---- ABI (cuda/include/matrix/MatrixNTT.cuh:9-10): idempotent w.r.t. the format tag -------------
extern "C" int gpu_matrix_ntt_all(GpuMatrix *mat) {
    ABI_GUARD_BEGIN
    if (!mat) return set_error("gpu_matrix_ntt_all: null matrix");
    if (mat->format == GPU_POLY_FORMAT_EVAL) return 0;
    if (ctx_activate(mat->ctx)) return 1;
    int rc = launch_ntt(mat->ctx, mat->data, matrix_polys(mat) * matrix_limbs(mat), mat->level + 1, false);
    if (rc) return rc;
    mat->format = GPU_POLY_FORMAT_EVAL;
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_intt_all(GpuMatrix *mat) {
    ABI_GUARD_BEGIN
    if (!mat) return set_error("gpu_matrix_intt_all: null matrix");
    if (mat->format == GPU_POLY_FORMAT_COEFF) return 0;
    if (ctx_activate(mat->ctx)) return 1;
    int rc = launch_ntt(mat->ctx, mat->data, matrix_polys(mat) * matrix_limbs(mat), mat->level + 1, true);
    if (rc) return rc;
    mat->format = GPU_POLY_FORMAT_COEFF;
    return 0;
    ABI_GUARD_END
}

// out <- INTT(lhs o scalar): the point-wise product by a resident 1x1 EVAL ring element rides in the
// inverse transform's load (extension; the reference runs gpu_matrix_mul_scalar then gpu_matrix_intt_all,
// two full HBM round trips).  `out` may be `lhs`.  Falls back to those two calls where no fused kernel exists.
extern "C" int gpupoly_matrix_mul_scalar_intt(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *scalar) {
    ABI_GUARD_BEGIN
    if (!scalar) return set_error("gpupoly_matrix_mul_scalar_intt: null scalar");
    if (matrix_check_same_shape(out, lhs, "gpupoly_matrix_mul_scalar_intt")) return 1;
    if (scalar->ctx != lhs->ctx || scalar->level != lhs->level)
        return set_error("gpupoly_matrix_mul_scalar_intt: context/level mismatch");
    if (scalar->rows != 1 || scalar->cols != 1) return set_error("gpupoly_matrix_mul_scalar_intt: scalar must be 1x1");
    if (lhs->format != GPU_POLY_FORMAT_EVAL || scalar->format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpupoly_matrix_mul_scalar_intt requires Eval format");
    GpuContext *ctx = out->ctx;
    if (matrix_polys(out) == 0) {
        out->format = GPU_POLY_FORMAT_COEFF;
        return 0;
    }
    if (ctx_activate(ctx)) return 1;
    int rc = -1;
    if (!ctx->wide)
        rc = launch_mul_intt_u32(ctx, static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(lhs->data),
                                 static_cast<const uint32_t *>(scalar->data), matrix_polys(out) * matrix_limbs(out),
                                 static_cast<uint32_t>(matrix_limbs(out)));
    if (rc < 0) {
        rc = gpu_matrix_mul_scalar(out, lhs, scalar);
        if (!rc) rc = gpu_matrix_intt_all(out);
        return rc;
    }
    if (rc) return rc;
    out->format = GPU_POLY_FORMAT_COEFF;
    return 0;
    ABI_GUARD_END
}
