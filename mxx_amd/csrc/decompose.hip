// decompose.hip — gadget matrices and base-2^b digit decomposition.
// Replaces cuda/src/matrix/MatrixDecompose.cu behind
// cuda/include/matrix/MatrixDecompose.cuh:26-37.
//
// Indexing (MatrixDecompose.cu:77-113, CPU twin
// src/matrix/dcrt_poly.rs:134-198,453-493): dpt = ceil(crt_bits/base_bits),
// k = dpt*(level+1); digit d of the limb-t residue of coefficient i of M[r,c]
// goes to row r*k + t*dpt + d, column c, coefficient i, replicated into every
// limb; the last digit of a tower keeps bits(q_t) - (dpt-1)*base_bits bits.
//
// One launch reads every source residue once and writes all its digits for all
// output limbs (the reference launches per source limb with a z-slice per digit
// after a full memset).  A constant polynomial is its own NTT, so the gadget
// matrices are written directly in EVAL form with no transform.
#include "common.h"
#include "modarith.h"

#include <algorithm>

static inline uint32_t host_bits(uint64_t v) { return v ? 64 - (uint32_t)__builtin_clzll(v) : 0; }

template <typename W>
__global__ void decompose_kernel(W *__restrict__ out, const W *__restrict__ src, const LimbConst *__restrict__ limbs,
                                 size_t src_polys, uint32_t src_cols, uint32_t L, uint32_t N, uint32_t towers,
                                 uint32_t dpt, uint32_t base_bits, size_t k) {
    // item = (src poly, tower, coefficient)
    const size_t idx = item_index();
    const size_t total = src_polys * towers * N;
    if (idx >= total) return;
    const uint32_t i = static_cast<uint32_t>(idx % N);
    const size_t pt = idx / N;
    const uint32_t t = static_cast<uint32_t>(pt % towers);
    const size_t p = pt / towers;
    const size_t r = p / src_cols, c = p - r * src_cols;
    const uint64_t residue = static_cast<uint64_t>(src[(p * L + t) * N + i]);
    const uint32_t src_bits = limbs[t].kbits;
    for (uint32_t d = 0; d < dpt; ++d) {
        const uint32_t shift = d * base_bits;
        uint64_t mask = 0;
        if (shift < src_bits) {
            const uint32_t rem = src_bits - shift;
            const uint32_t db = base_bits < rem ? base_bits : rem;
            mask = db >= 64 ? ~0ull : ((1ull << db) - 1);
        }
        const uint64_t digit = shift >= 64 ? 0 : ((residue >> shift) & mask);
        const size_t orow = r * k + static_cast<size_t>(t) * dpt + d;
        const size_t opoly = orow * src_cols + c;
        for (uint32_t l = 0; l < L; ++l) {
            const uint64_t ql = limbs[l].q;
            out[(opoly * L + l) * N + i] = static_cast<W>(digit >= ql ? digit % ql : digit);
        }
    }
}

// G = I_size (x) g written in EVAL form (all slots of a constant poly are equal)
template <typename W>
__global__ void fill_gadget_kernel(W *__restrict__ out, const LimbConst *__restrict__ limbs, size_t rows, size_t cols,
                                   uint32_t L, uint32_t N, uint32_t dpt, size_t k, uint32_t base_bits, int small) {
    const size_t idx = item_index();
    const size_t total = rows * cols * L * N;
    if (idx >= total) return;
    const size_t pl = idx / N;
    const uint32_t l = static_cast<uint32_t>(pl % L);
    const size_t p = pl / L;
    const size_t r = p / cols, c = p - r * cols;
    W value = 0;
    const size_t start = r * k;
    if (c >= start && c < start + k) {
        const size_t local = c - start;
        const uint32_t tower = static_cast<uint32_t>(local / dpt), digit = static_cast<uint32_t>(local % dpt);
        if (small || tower == l) {
            const LimbConst lc = limbs[l];
            const uint64_t q = lc.q;
            // base^digit mod q by repeated multiplication (digit < dpt <= 64)
            const uint64_t base = (base_bits >= 64 ? 0 : (1ull << base_bits)) % q;
            uint64_t v = 1 % q;
            for (uint32_t e = 0; e < digit; ++e) v = static_cast<uint64_t>((static_cast<u128_t>(v) * base) % q);
            value = static_cast<W>(v);
        }
    }
    out[idx] = value;
}

// out[local_row, src_row] = scalar_by_digit[0, digit]; everything else was zeroed
template <typename W>
__global__ void identity_chunk_kernel(W *__restrict__ out, const W *__restrict__ src, size_t size, size_t chunk_idx,
                                      size_t chunk_count, size_t words_per_poly) {
    const size_t local_row = blockIdx.y;
    const size_t global_row = chunk_idx * size + local_row;
    const size_t src_row = global_row / chunk_count;
    const size_t digit = global_row - src_row * chunk_count;
    if (src_row >= size) return;
    const W *s = src + digit * words_per_poly;
    W *d = out + (local_row * size + src_row) * words_per_poly;
    for (size_t w = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < words_per_poly;
         w += static_cast<size_t>(gridDim.x) * blockDim.x)
        d[w] = s[w];
}

static int fill_gadget_impl(GpuMatrix *out, uint32_t base_bits, bool small) {
    if (!out) return set_error("gpu_matrix_fill_gadget: null matrix");
    if (base_bits == 0 || base_bits >= 63) return set_error("gpu_matrix_fill_gadget: invalid base_bits");
    GpuContext *ctx = out->ctx;
    const uint32_t dpt = (ctx->crt_bits + base_bits - 1) / base_bits;
    const size_t L = matrix_limbs(out);
    const size_t k = small ? dpt : static_cast<size_t>(dpt) * L;
    if (out->cols != out->rows * k) return set_error("gpu_matrix_fill_gadget: output must be size x size*log_base_q");
    out->format = GPU_POLY_FORMAT_EVAL;
    const size_t total = matrix_words(out);
    if (total == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    const dim3 blocks = item_grid(total, 256);
    if (ctx->wide)
        MXX_LAUNCH(fill_gadget_kernel<uint64_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(out->data), ctx->d_limbs, out->rows, out->cols, (uint32_t)L,
                           (uint32_t)ctx->N, dpt, k, base_bits, small ? 1 : 0);
    else
        MXX_LAUNCH(fill_gadget_kernel<uint32_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(out->data), ctx->d_limbs, out->rows, out->cols, (uint32_t)L,
                           (uint32_t)ctx->N, dpt, k, base_bits, small ? 1 : 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int decompose_impl(const GpuMatrix *src, uint32_t base_bits, GpuMatrix *out, bool small) {
    if (!src || !out) return set_error("gpu_matrix_decompose_base: null matrix");
    if (base_bits == 0) return set_error("base_bits must be non-zero in gpu_matrix_decompose_base");
    if (src->ctx != out->ctx || src->level != out->level)
        return set_error("context mismatch in gpu_matrix_decompose_base");
    if (src == out) return set_error("gpu_matrix_decompose_base: output must not alias the source");
    GpuContext *ctx = src->ctx;
    const int requested = out->format;
    const size_t L = matrix_limbs(src);
    const uint32_t dpt = (ctx->crt_bits + base_bits - 1) / base_bits;
    const size_t k = small ? dpt : static_cast<size_t>(dpt) * L;
    if (out->rows != src->rows * k || out->cols != src->cols)
        return set_error("output size mismatch in gpu_matrix_decompose_base");
    const size_t polys = matrix_polys(src);
    if (polys == 0) {
        out->format = GPU_POLY_FORMAT_EVAL;
        return 0;
    }
    if (ctx_activate(ctx)) return 1;
    // digits are taken from coefficient-domain residues: INTT a private copy if needed
    const void *coeff = src->data;
    CtxBlock tmp_block(ctx);  // back to the cache at scope exit (stream-ordered behind its readers), error paths included
    if (src->format == GPU_POLY_FORMAT_EVAL) {
        if (tmp_block.alloc(src->bytes)) return 1;
        int rc = ctx->wide ? -1 : launch_intt_oop_u32(ctx, static_cast<uint32_t *>(tmp_block.ptr), static_cast<const uint32_t *>(src->data),
                                                      polys * L, static_cast<uint32_t>(L));
        if (rc < 0) {  // no out-of-place transform for this context: copy, then in place
            MXX_TRACED_COPY("copy (device to device)", ctx->stream, 2.0 * src->bytes,
                            HIP_TRY(hipMemcpyAsync(tmp_block.ptr, src->data, src->bytes, hipMemcpyDeviceToDevice, ctx->stream)));
            rc = launch_ntt(ctx, tmp_block.ptr, polys * L, static_cast<int>(L), true);
        }
        if (rc) return rc;
        coeff = tmp_block.ptr;
    }
    const uint32_t towers = small ? 1u : static_cast<uint32_t>(L);
    if (requested == GPU_POLY_FORMAT_EVAL) {
        // digits generated inside the forward transform's load: no COEFF digit matrix is written
        const int frc = ctx->wide ? launch_ntt_digits_u64(ctx, static_cast<uint64_t *>(out->data), static_cast<const uint64_t *>(coeff),
                                                          matrix_polys(out) * L, static_cast<uint32_t>(L), (uint32_t)src->cols,
                                                          towers, dpt, base_bits, k)
                                  : launch_ntt_digits_u32(ctx, static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(coeff),
                                                          matrix_polys(out) * L, static_cast<uint32_t>(L), (uint32_t)src->cols,
                                                          towers, dpt, base_bits, k);
        if (frc >= 0) {
            if (frc == 0) out->format = GPU_POLY_FORMAT_EVAL;
            return frc;
        }
    }
    const size_t total = polys * towers * static_cast<size_t>(ctx->N);
    const dim3 blocks = item_grid(total, 256);
    if (ctx->wide)
        MXX_LAUNCH(decompose_kernel<uint64_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(out->data), static_cast<const uint64_t *>(coeff), ctx->d_limbs, polys,
                           (uint32_t)src->cols, (uint32_t)L, (uint32_t)ctx->N, towers, dpt, base_bits, k);
    else
        MXX_LAUNCH(decompose_kernel<uint32_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(coeff), ctx->d_limbs, polys,
                           (uint32_t)src->cols, (uint32_t)L, (uint32_t)ctx->N, towers, dpt, base_bits, k);
    HIP_TRY(hipGetLastError());
    out->format = GPU_POLY_FORMAT_COEFF;
    if (requested == GPU_POLY_FORMAT_EVAL) {
        // output honours the format it was created with (MatrixDecompose.cu:910-914,1318-1328)
        int rc = launch_ntt(ctx, out->data, matrix_polys(out) * L, static_cast<int>(L), false);
        if (rc) return rc;
        out->format = GPU_POLY_FORMAT_EVAL;
    }
    return 0;
}

extern "C" int gpu_matrix_fill_gadget(GpuMatrix *out, uint32_t base_bits) {
    ABI_GUARD_BEGIN
    return fill_gadget_impl(out, base_bits, false);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_fill_small_gadget(GpuMatrix *out, uint32_t base_bits) {
    ABI_GUARD_BEGIN
    return fill_gadget_impl(out, base_bits, true);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_decompose_base(const GpuMatrix *src, uint32_t base_bits, GpuMatrix *out) {
    ABI_GUARD_BEGIN
    return decompose_impl(src, base_bits, out, false);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_decompose_base_small(const GpuMatrix *src, uint32_t base_bits, GpuMatrix *out) {
    ABI_GUARD_BEGIN
    return decompose_impl(src, base_bits, out, true);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_fill_small_decomposed_identity_chunk(GpuMatrix *out, const GpuMatrix *scalar_by_digit,
                                                               size_t chunk_idx) {
    ABI_GUARD_BEGIN
    if (!out || !scalar_by_digit) return set_error("invalid gpu_matrix_fill_small_decomposed_identity_chunk arguments");
    if (out->ctx != scalar_by_digit->ctx || out->level != scalar_by_digit->level)
        return set_error("context mismatch in gpu_matrix_fill_small_decomposed_identity_chunk");
    if (out->rows != out->cols)
        return set_error("output must be square in gpu_matrix_fill_small_decomposed_identity_chunk");
    if (scalar_by_digit->rows != 1 || scalar_by_digit->cols == 0)
        return set_error("scalar_by_digit must be 1 x chunk_count in gpu_matrix_fill_small_decomposed_identity_chunk");
    if (out->format != scalar_by_digit->format)
        return set_error("format mismatch in gpu_matrix_fill_small_decomposed_identity_chunk");
    const size_t size = out->rows, chunk_count = scalar_by_digit->cols;
    if (chunk_idx >= chunk_count)
        return set_error("chunk_idx out of range in gpu_matrix_fill_small_decomposed_identity_chunk");
    if (size == 0) return 0;
    if (size > 65535) return set_error("gpu_matrix_fill_small_decomposed_identity_chunk: size too large");
    GpuContext *ctx = out->ctx;
    if (ctx_activate(ctx)) return 1;
    HIP_TRY(hipMemsetAsync(out->data, 0, out->bytes, ctx->stream));
    const size_t wpp = matrix_limbs(out) * static_cast<size_t>(ctx->N);
    const unsigned gx = static_cast<unsigned>(std::min<size_t>((wpp + 255) / 256, 64));
    dim3 grid(gx, static_cast<unsigned>(size));
    if (ctx->wide)
        MXX_LAUNCH(identity_chunk_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(out->data), static_cast<const uint64_t *>(scalar_by_digit->data),
                           size, chunk_idx, chunk_count, wpp);
    else
        MXX_LAUNCH(identity_chunk_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(scalar_by_digit->data),
                           size, chunk_idx, chunk_count, wpp);
    HIP_TRY(hipGetLastError());
    return 0;
    ABI_GUARD_END
}

// Extension: G^-1 of a freshly sampled matrix in one call.  The reference samples (coefficients -> NTT), then
// decomposes (INTT of a copy -> digits -> NTT): src/sampler/gpu.rs:91-115 (`sample_hash_decomposed`,
// `sample_hash_small_decomposed`).  Here the samples stay coefficients and go straight into the digit transform:
// the source's NTT, its copy and its INTT are never run.  `out` is (rows * k) x cols, created EVAL or COEFF like
// the output of gpu_matrix_decompose_base; the samples are those of gpu_matrix_sample_distribution(rows x cols).
extern "C" int gpupoly_matrix_sample_decomposed(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed,
                                                uint32_t base_bits, int small) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpupoly_matrix_sample_decomposed: null matrix");
    if (base_bits == 0) return set_error("base_bits must be non-zero in gpupoly_matrix_sample_decomposed");
    GpuContext *ctx = out->ctx;
    const size_t L = matrix_limbs(out);
    const uint32_t dpt = (ctx->crt_bits + base_bits - 1) / base_bits;
    const size_t k = small ? dpt : static_cast<size_t>(dpt) * L;
    if (out->rows % k) return set_error("gpupoly_matrix_sample_decomposed: output rows must be a multiple of the digit count");
    GpuMatrix src;
    src.ctx = ctx;
    src.level = out->level;
    src.rows = out->rows / k;
    src.cols = out->cols;
    src.bytes = src.rows * src.cols * L * static_cast<size_t>(ctx->N) * static_cast<size_t>(ctx->word_bytes);
    if (matrix_polys(&src) == 0) {
        out->format = GPU_POLY_FORMAT_EVAL;
        return 0;
    }
    if (ctx_activate(ctx)) return 1;
    CtxBlock block(ctx);
    if (block.alloc(src.bytes)) return 1;
    src.data = block.ptr;
    int rc = sample_impl(&src, dist_type, sigma, seed, src.cols, 0, true);
    if (rc == 0) rc = decompose_impl(&src, base_bits, out, small != 0);
    return rc;
    ABI_GUARD_END
}
