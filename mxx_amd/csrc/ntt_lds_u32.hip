// ntt_lds_u32.hip — instantiations of the lazy LDS NTT for 32-bit residue words.
#include "ntt14.h"

#include <atomic>
#include <cstdlib>

typedef uint32_t W;
#include "ntt_lds_dispatch.inc"

int launch_ntt_lds_u32(GpuContext *ctx, uint32_t *data, size_t vectors, uint32_t L, bool inverse) {
    return dispatch_ntt_lds(ctx, data, vectors, L, inverse);
}

int launch_mul_intt_u32(GpuContext *ctx, uint32_t *out, const uint32_t *in, const uint32_t *w, size_t vectors, uint32_t L) {
    return launch_mul_intt(ctx, out, in, w, vectors, L);
}

// decompose + forward NTT in one pass (ntt14.h, fwd_digits_kernel); -1: not available for this
// context / path override, the caller then runs the digit kernel and the transform separately
int launch_ntt_digits_u32(GpuContext *ctx, uint32_t *out, const uint32_t *coeff, size_t out_vectors, uint32_t L,
                          uint32_t src_cols, uint32_t towers, uint32_t dpt, uint32_t base_bits, size_t k) {
    const EnvSwitches &env = ctx->env;
    if (ctx->logN != 14 || !ctx->lazy_ok || env.ntt14 == 1 || env.ntt_path > 1 || !env.decompose_fused ||
        out_vectors > 0x7fffffffull || k >> 32)
        return -1;
    hipLaunchKernelGGL(ntt14::fwd_digits_kernel<W>, dim3(static_cast<unsigned>(out_vectors)), dim3(ntt14::T),
                       ntt14::lds_bytes(sizeof(W)), ctx->stream, out, coeff, static_cast<const TwPair<W> *>(ctx->d_tw2_fwd),
                       ctx->d_limbs, L, src_cols, towers, dpt, base_bits, static_cast<uint32_t>(k));
    HIP_TRY(hipGetLastError());
    return 0;
}
