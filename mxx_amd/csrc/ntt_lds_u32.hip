// ntt_lds_u32.hip — instantiations of the lazy LDS NTT for 32-bit residue words.
#include "ntt14.h"

#include <algorithm>
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

// out = INTT(in), `in` left untouched (ntt14.h, inv_kernel reading from `in`): the decompose paths need the coefficients
// of an EVAL source that must stay EVAL - a device copy followed by the in-place transform moves the matrix three
// times, this moves it twice.  -1: the grouped 2^14 kernel does not run for this context / path override, the caller
// then copies and transforms in place.
int launch_intt_oop_u32(GpuContext *ctx, uint32_t *out, const uint32_t *in, size_t vectors, uint32_t L) {
    const EnvSwitches &env = ctx->env;
    if (ctx->logN != 14 || !(ctx->lazy_ok || ctx->tight_ok) || env.ntt14 == 1 || env.ntt_path > 1) return -1;
    dim3 grid;
    if (vectors == 0 || vectors > 0x7fffffffull || !ntt14_grid(vectors, L, grid)) return -1;
    MXX_TRACE_BYTES(2.0 * vectors * ntt14::N * sizeof(W));
    return ctx->lazy_ok ? launch_ntt14<false>(ctx, out, vectors, L, true, in) : launch_ntt14<true>(ctx, out, vectors, L, true, in);
}

// out = NTT(src) + add in one pass (ntt14.h, fwd_add_kernel); -1: no fused kernel for this context / path override,
// the caller then copies, transforms in place and adds
int launch_ntt_add_u32(GpuContext *ctx, uint32_t *out, const uint32_t *src, const uint32_t *add, size_t vectors, uint32_t L) {
    const EnvSwitches &env = ctx->env;
    if (ctx->logN != 14 || !(ctx->lazy_ok || ctx->tight_ok) || env.ntt14 == 1 || env.ntt_path > 1) return -1;
    dim3 grid, block(ntt14::T);
    if (vectors > 0x7fffffffull || !ntt14_grid(vectors, L, grid)) return -1;
    const size_t lds = ntt14::lds_bytes(sizeof(W));
    const TwPair<W> *tw = static_cast<const TwPair<W> *>(ctx->d_tw2_fwd);
    MXX_TRACE_BYTES(3.0 * vectors * ntt14::N * sizeof(W));  // coefficients + addend read, the sum's transform written
    if (ctx->lazy_ok) MXX_LAUNCH((ntt14::fwd_add_kernel<W, false>), grid, block, lds, ctx->stream, out, src, add, tw, ctx->d_limbs, L);
    else MXX_LAUNCH((ntt14::fwd_add_kernel<W, true>), grid, block, lds, ctx->stream, out, src, add, tw, ctx->d_limbs, L);
    HIP_TRY(hipGetLastError());
    return 0;
}

// decompose + forward NTT in one pass (ntt14.h, fwd_digits_kernel); -1: not available for this
// context / path override, the caller then runs the digit kernel and the transform separately
int launch_ntt_digits_u32(GpuContext *ctx, uint32_t *out, const uint32_t *coeff, size_t out_vectors, uint32_t L,
                          uint32_t src_cols, uint32_t towers, uint32_t dpt, uint32_t base_bits, size_t k) {
    const EnvSwitches &env = ctx->env;
    const bool tight = !ctx->lazy_ok;
    if (!(ctx->lazy_ok || ctx->tight_ok) || env.ntt14 == 1 || env.ntt_path > 1 || !env.decompose_fused ||
        out_vectors > 0x7fffffffull || k >> 32)
        return -1;
    if (k == 0 || src_cols == 0 || out_vectors % (k * src_cols * L) != 0) return -1;
    const size_t src_rows = out_vectors / (k * src_cols * L);
    if (src_rows > 65535 || k > 65535 || static_cast<uint64_t>(src_cols) * L > 0x7fffffffull) return -1;
    // can a digit reach an output modulus?  (digits are below 2^min(base_bits, bits of the widest limb))
    const uint32_t digit_bits = std::min<uint32_t>(base_bits, ctx->crt_bits);
    uint64_t min_q = ~0ull;
    for (uint32_t l = 0; l < L; ++l) min_q = std::min<uint64_t>(min_q, ctx->moduli[l]);
    const bool reduce = digit_bits >= 63 || ((1ull << digit_bits) - 1) >= min_q;
    (void)towers;
    // SURVEY 8d decompose: (r c + r k c) n L w - the source read once, the digit matrix written once
    MXX_TRACE_BYTES((static_cast<double>(src_rows) * src_cols * L + static_cast<double>(out_vectors)) * ctx->N * sizeof(W));
    if (ctx->logN != 14) return dispatch_ntt_digits(ctx, out, coeff, L, src_cols, src_rows, dpt, base_bits, k, reduce);
    const dim3 grid(8u * L * ((src_cols + 7u) / 8u), static_cast<unsigned>(k), static_cast<unsigned>(src_rows));
    const dim3 block(ntt14::T);
    const size_t lds = ntt14::lds_bytes(sizeof(W));
    const TwPair<W> *tw = static_cast<const TwPair<W> *>(ctx->d_tw2_fwd);
    const uint32_t k32 = static_cast<uint32_t>(k);
    const bool nts = out_vectors * sizeof(W) * ntt14::N >= (size_t(1) << 29);  // outputs that fit the Infinity Cache stay cacheable for their consumer ((1 x 64) G^-1(4 x 4): 180 -> 168 us); from 0.5 GB the hint wins (8 x 8: 442 -> 390 us)
#define MXX_DIGITS(RED, TGT)                                                                                                     \
    do {                                                                                                                         \
        if (nts)                                                                                                                 \
            MXX_LAUNCH((ntt14::fwd_digits_kernel<W, RED, TGT, true>), grid, block, lds, ctx->stream, out, coeff, tw,      \
                               ctx->d_limbs, L, src_cols, towers, dpt, base_bits, k32);                                           \
        else                                                                                                                     \
            MXX_LAUNCH((ntt14::fwd_digits_kernel<W, RED, TGT, false>), grid, block, lds, ctx->stream, out, coeff, tw,     \
                               ctx->d_limbs, L, src_cols, towers, dpt, base_bits, k32);                                           \
    } while (0)
    if (reduce) {
        if (tight) MXX_DIGITS(true, true);
        else MXX_DIGITS(true, false);
    } else {
        if (tight) MXX_DIGITS(false, true);
        else MXX_DIGITS(false, false);
    }
#undef MXX_DIGITS
    HIP_TRY(hipGetLastError());
    return 0;
}
