// ntt_lds_u64.hip — instantiations of the lazy LDS NTT for 64-bit residue words.
#include "ntt14.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

typedef uint64_t W;
#include "ntt_lds_dispatch.inc"

int launch_ntt_lds_u64(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse) {
    return dispatch_ntt_lds(ctx, data, vectors, L, inverse);
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
    return dispatch_ntt_digits(ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce);
}
