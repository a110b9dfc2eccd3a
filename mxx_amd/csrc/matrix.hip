// matrix.hip — matrix storage, block copies and host<->device RNS marshalling.
// Replaces cuda/src/matrix/MatrixData.cu and the RNS-batch half of MatrixSerde.cu
// (reference ABI: cuda/include/matrix/MatrixData.cuh:10-27,
//  cuda/include/matrix/MatrixSerde.cuh:10-33).
//
// Device layout: one allocation per matrix, words [poly][limb][N] with
// poly = row*cols + col.  A poly (all limbs) is contiguous, so a row segment
// of a sub-block is one contiguous run and the host wire layout
// ([poly][limb][N] u64, src/poly/dcrt/gpu.rs:758-788) maps 1:1 onto it.
#include "common.h"
#include "modarith.h"

#include <algorithm>

// ---- kernels ------------------------------------------------------------------------

// host-order u64 staging -> device words (narrowing when W = uint32_t)
template <typename W>
__global__ void unpack_rns_kernel(const uint64_t *__restrict__ src, W *__restrict__ dst, size_t words_per_poly_dst,
                                  size_t words_per_poly_src, size_t total) {
    size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; idx < total; idx += stride) {
        size_t poly = idx / words_per_poly_dst;
        size_t w = idx - poly * words_per_poly_dst;
        dst[idx] = static_cast<W>(src[poly * words_per_poly_src + w]);
    }
}

template <typename W>
__global__ void pack_rns_kernel(const W *__restrict__ src, uint64_t *__restrict__ dst, size_t words_per_poly_src,
                                size_t words_per_poly_dst, size_t total) {
    size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; idx < total; idx += stride) {
        size_t poly = idx / words_per_poly_src;
        size_t w = idx - poly * words_per_poly_src;
        dst[poly * words_per_poly_dst + w] = static_cast<uint64_t>(src[idx]);
    }
}

// constant coefficient (index 0) of every limb of every poly -> u64 words
template <typename W>
__global__ void const_coeff_kernel(const W *__restrict__ src, uint64_t *__restrict__ dst, size_t polys, size_t limbs,
                                   size_t N, size_t words_per_poly_dst) {
    size_t idx = item_index();
    if (idx >= polys * limbs) return;
    size_t poly = idx / limbs, l = idx - poly * limbs;
    dst[poly * words_per_poly_dst + l] = static_cast<uint64_t>(src[(poly * limbs + l) * N]);
}

// rectangular block copy / accumulate; one blockIdx.y per (row, col) entry of the block
template <typename W, bool ADD>
__global__ void block_rect_kernel(W *__restrict__ dst, const W *__restrict__ src, const LimbConst *__restrict__ limbs,
                                  size_t dst_cols, size_t src_cols, size_t dst_row, size_t dst_col, size_t src_row,
                                  size_t src_col, size_t cols, size_t words_per_poly, uint32_t N) {
    size_t entry = blockIdx.y;
    size_t r = entry / cols, c = entry - r * cols;
    const W *s = src + ((src_row + r) * src_cols + (src_col + c)) * words_per_poly;
    W *d = dst + ((dst_row + r) * dst_cols + (dst_col + c)) * words_per_poly;
    for (size_t w = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < words_per_poly;
         w += static_cast<size_t>(gridDim.x) * blockDim.x) {
        if (ADD) {
            W q = static_cast<W>(limbs[w / N].q);
            d[w] = add_mod<W>(d[w], s[w], q);
        } else {
            d[w] = s[w];
        }
    }
}

// out[c][r] = src[r][c], whole polynomials (all limbs contiguous): blockIdx.y = destination entry, 16 bytes per lane
__global__ void transpose_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t src_rows, size_t src_cols,
                                 size_t vec_per_poly) {
    const size_t entry = static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y;  // destination index c * src_rows + r
    if (entry >= src_rows * src_cols) return;
    const size_t c = entry / src_rows, r = entry - c * src_rows;
    const uint4 *s = src + (r * src_cols + c) * vec_per_poly;
    uint4 *d = dst + entry * vec_per_poly;
    for (size_t v = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < vec_per_poly;
         v += static_cast<size_t>(gridDim.x) * blockDim.x)
        d[v] = s[v];
}

// int64 [poly][N] -> residues in every limb of `out`
template <typename W>
__global__ void scatter_i64_kernel(W *__restrict__ dst, const int64_t *__restrict__ vals,
                                   const LimbConst *__restrict__ limbs, size_t polys, uint32_t L, uint32_t N) {
    size_t idx = item_index();
    size_t total = polys * N;
    if (idx >= total) return;
    size_t poly = idx / N;
    uint32_t i = static_cast<uint32_t>(idx - poly * N);
    int64_t v = vals[idx];
    if constexpr (sizeof(W) == 4) {
        // |v| below 2^32 in every lane of the wave (perturbation samples: sigma_large ~ 1e8 against 24-bit moduli, so nearly
        // every sample needs a real reduction and the generic routine's 64-bit path was this kernel's whole cost - 344
        // instructions per element for ten limbs): one 32-bit Barrett step per limb, the estimate floor(m mu32 / 2^32) is at
        // most 3 short for a modulus below 2^30, two min-style corrections, then the sign
        const bool neg = v < 0;
        const uint64_t mag = neg ? 0ull - static_cast<uint64_t>(v) : static_cast<uint64_t>(v);
        if (__all((mag >> 32) == 0)) {
            const uint32_t m = static_cast<uint32_t>(mag);
            for (uint32_t l = 0; l < L; ++l) {
                const LimbConst &lc = limbs[l];
                W *out = dst + (poly * L + l) * N + i;
                if (lc.kbits <= 30) {
                    const uint32_t q = static_cast<uint32_t>(lc.q);
                    uint32_t r = m - __umulhi(m, static_cast<uint32_t>(lc.mu32)) * q;  // in [0, 4q)
                    r = min(r, r - 2u * q);
                    r = min(r, r - q);
                    *out = (neg && r != 0u) ? q - r : r;
                } else {
                    *out = signed_to_residue_mu<W>(v, lc.q, lc.mu64);
                }
            }
            return;
        }
    }
    for (uint32_t l = 0; l < L; ++l) dst[(poly * L + l) * N + i] = signed_to_residue_mu<W>(v, limbs[l].q, limbs[l].mu64);
}

// ---- helpers ---------------------------------------------------------------------------
int matrix_check_same_shape(const GpuMatrix *a, const GpuMatrix *b, const char *who) {
    if (!a || !b) return set_error(std::string(who) + ": null matrix");
    if (a->ctx != b->ctx) return set_error(std::string(who) + ": context mismatch");
    if (a->level != b->level) return set_error(std::string(who) + ": level mismatch");
    if (a->rows != b->rows || a->cols != b->cols) return set_error(std::string(who) + ": shape mismatch");
    return 0;
}

static inline unsigned grid_for(size_t total, unsigned threads, unsigned cap = 8192) {
    size_t blocks = (total + threads - 1) / threads;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return static_cast<unsigned>(blocks);
}

int launch_scatter_i64(GpuMatrix *out, const int64_t *vals) {
    GpuContext *ctx = out->ctx;
    size_t polys = matrix_polys(out);
    if (polys == 0) return 0;
    size_t total = polys * ctx->N;
    const dim3 blocks = item_grid(total, 256);
    uint32_t L = static_cast<uint32_t>(matrix_limbs(out));
    MXX_TRACE_BYTES(static_cast<double>(total) * 8 + out->bytes);  // int64 staging read, residues of every limb written
    if (ctx->wide)
        MXX_LAUNCH(scatter_i64_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(out->data), vals, ctx->d_limbs, polys, L, (uint32_t)ctx->N);
    else
        MXX_LAUNCH(scatter_i64_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(out->data), vals, ctx->d_limbs, polys, L, (uint32_t)ctx->N);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_copy_block(GpuMatrix *out, const GpuMatrix *src, size_t dst_row, size_t dst_col, size_t src_row,
                      size_t src_col, size_t rows, size_t cols, bool add) {
    const char *who = add ? "gpu_matrix_add_block" : "gpu_matrix_copy_block";
    if (!out || !src) return set_error(std::string(who) + ": null matrix");
    if (out->ctx != src->ctx) return set_error(std::string(who) + ": context mismatch");
    if (out->level != src->level) return set_error(std::string(who) + ": level mismatch");
    if (src_row + rows > src->rows || src_col + cols > src->cols)
        return set_error(std::string(who) + ": source block out of bounds");
    if (dst_row + rows > out->rows || dst_col + cols > out->cols)
        return set_error(std::string(who) + ": destination block out of bounds");
    // quirk kept from the reference: the whole destination is retagged with the
    // source's format even for a partial block (MatrixData.cu:519,531,558)
    out->format = src->format;
    if (rows == 0 || cols == 0) return 0;
    GpuContext *ctx = out->ctx;
    if (ctx_activate(ctx)) return 1;
    size_t wpp = matrix_limbs(out) * static_cast<size_t>(ctx->N);
    if (!add) {
        size_t wb = static_cast<size_t>(ctx->word_bytes);
        const char *s = static_cast<const char *>(src->data) + (src_row * src->cols + src_col) * wpp * wb;
        char *d = static_cast<char *>(out->data) + (dst_row * out->cols + dst_col) * wpp * wb;
        if (out == src) {
            // overlapping self-copy: go through the kernel only when disjoint is not guaranteed
            // (hipMemcpy2D has undefined overlap semantics); use a temp
            CtxBlock tmp(ctx);  // back to the cache at scope exit, error paths included
            size_t row_bytes = cols * wpp * wb;
            if (tmp.alloc(rows * row_bytes)) return 1;
            MXX_TRACED_COPY("copy_block (2-D runtime copy)", ctx->stream, 2.0 * rows * row_bytes,
                            HIP_TRY(hipMemcpy2DAsync(tmp.ptr, row_bytes, s, src->cols * wpp * wb, row_bytes, rows,
                                                     hipMemcpyDeviceToDevice, ctx->stream)));
            MXX_TRACED_COPY("copy_block (2-D runtime copy)", ctx->stream, 2.0 * rows * row_bytes,
                            HIP_TRY(hipMemcpy2DAsync(d, out->cols * wpp * wb, tmp.ptr, row_bytes, row_bytes, rows,
                                                     hipMemcpyDeviceToDevice, ctx->stream)));
            return 0;
        }
        if (rows == 1 || (cols == src->cols && cols == out->cols)) {
            MXX_TRACED_COPY("copy_block (runtime copy)", ctx->stream, 2.0 * rows * cols * wpp * wb,
                            HIP_TRY(hipMemcpyAsync(d, s, rows * cols * wpp * wb, hipMemcpyDeviceToDevice, ctx->stream)));
        } else {
            MXX_TRACED_COPY("copy_block (2-D runtime copy)", ctx->stream, 2.0 * rows * cols * wpp * wb,
                            HIP_TRY(hipMemcpy2DAsync(d, out->cols * wpp * wb, s, src->cols * wpp * wb, cols * wpp * wb, rows,
                                                     hipMemcpyDeviceToDevice, ctx->stream)));
        }
        return 0;
    }
    size_t entries = rows * cols;
    // blockIdx.y is limited to 65535: chunk over row groups
    size_t rows_per_launch = std::max<size_t>(1, 65535 / cols);
    if (cols > 65535) return set_error(std::string(who) + ": block too wide");
    (void)entries;
    unsigned gx = grid_for(wpp, 256, 64);
    for (size_t r0 = 0; r0 < rows; r0 += rows_per_launch) {
        size_t rr = std::min(rows_per_launch, rows - r0);
        dim3 grid(gx, static_cast<unsigned>(rr * cols));
        MXX_TRACE_BYTES(3.0 * rr * cols * wpp * ctx->word_bytes);  // destination block read + written, source block read
        if (ctx->wide)
            MXX_LAUNCH((block_rect_kernel<uint64_t, true>), grid, dim3(256), 0, ctx->stream,
                               static_cast<uint64_t *>(out->data), static_cast<const uint64_t *>(src->data),
                               ctx->d_limbs, out->cols, src->cols, dst_row + r0, dst_col, src_row + r0, src_col, cols,
                               wpp, (uint32_t)ctx->N);
        else
            MXX_LAUNCH((block_rect_kernel<uint32_t, true>), grid, dim3(256), 0, ctx->stream,
                               static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(src->data),
                               ctx->d_limbs, out->cols, src->cols, dst_row + r0, dst_col, src_row + r0, src_col, cols,
                               wpp, (uint32_t)ctx->N);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// ---- ABI: storage -------------------------------------------------------------------------
extern "C" int gpu_matrix_create(GpuContext *ctx, int level, size_t rows, size_t cols, int format, GpuMatrix **out) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpu_matrix_create: null out");
    *out = nullptr;
    if (!ctx) return set_error("gpu_matrix_create: null context");
    if (level < 0 || level > ctx->level) return set_error("gpu_matrix_create: invalid level");
    if (format != GPU_POLY_FORMAT_COEFF && format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpu_matrix_create: invalid format");
    if (ctx_activate(ctx)) return 1;
    GpuMatrix *m = new GpuMatrix();
    m->ctx = ctx;
    m->level = level;
    m->rows = rows;
    m->cols = cols;
    m->format = format;
    m->bytes = matrix_words(m) * static_cast<size_t>(ctx->word_bytes);
    if (m->bytes) {
        if (ctx_alloc(ctx, m->bytes, &m->data)) {
            delete m;
            return 1;
        }
    }
    *out = m;
    return 0;
    ABI_GUARD_END
}

// Extension: a matrix object over rows [row, row + rows) of `m` - a contiguous block of the row-major layout - that
// shares m's storage: operands of a product or a sum without the slice's copy.  The view carries its own format tag
// (initialised from m's); it must be destroyed before m, and m must not be resized meanwhile (nothing here resizes).
extern "C" int gpupoly_matrix_row_view(GpuMatrix *m, size_t row, size_t rows, GpuMatrix **out_view) {
    ABI_GUARD_BEGIN
    if (!m || !out_view) return set_error("gpupoly_matrix_row_view: null argument");
    if (row > m->rows || rows > m->rows - row) return set_error("gpupoly_matrix_row_view: row block out of range");
    const size_t poly_bytes = matrix_limbs(m) * static_cast<size_t>(m->ctx->N) * m->ctx->word_bytes;
    GpuMatrix *v = new GpuMatrix(*m);
    v->rows = rows;
    v->data = rows && m->cols ? static_cast<char *>(m->data) + row * m->cols * poly_bytes : nullptr;
    v->bytes = rows * m->cols * poly_bytes;
    v->borrowed = true;
    *out_view = v;
    return 0;
    ABI_GUARD_END
}

extern "C" void gpu_matrix_destroy(GpuMatrix *mat) {
    if (!mat) return;
    if (mat->data && !mat->borrowed) {
        (void)hipSetDevice(mat->ctx->device);
        ctx_free(mat->ctx, mat->data);  // stream-ordered: in-flight kernels finish first
    }
    delete mat;
}

extern "C" int gpu_matrix_copy(GpuMatrix *dst, const GpuMatrix *src) {
    ABI_GUARD_BEGIN
    if (matrix_check_same_shape(dst, src, "gpu_matrix_copy")) return 1;
    dst->format = src->format;
    if (dst->bytes == 0 || dst == src) return 0;
    if (ctx_activate(dst->ctx)) return 1;
    MXX_TRACED_COPY("copy (device to device)", dst->ctx->stream, 2.0 * dst->bytes,
                    HIP_TRY(hipMemcpyAsync(dst->data, src->data, dst->bytes, hipMemcpyDeviceToDevice, dst->ctx->stream)));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_copy_block(GpuMatrix *out, const GpuMatrix *src, size_t dst_row, size_t dst_col,
                                     size_t src_row, size_t src_col, size_t rows, size_t cols) {
    ABI_GUARD_BEGIN
    return launch_copy_block(out, src, dst_row, dst_col, src_row, src_col, rows, cols, false);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_add_block(GpuMatrix *out, const GpuMatrix *src, size_t dst_row, size_t dst_col,
                                    size_t src_row, size_t src_col, size_t rows, size_t cols) {
    ABI_GUARD_BEGIN
    return launch_copy_block(out, src, dst_row, dst_col, src_row, src_col, rows, cols, true);
    ABI_GUARD_END
}

// Transpose in ONE launch (extension).  The reference's wrapper issues rows x cols single-polynomial
// gpu_matrix_copy_block calls (src/matrix/gpu_dcrt_poly.rs:1190-1199): 3600 launches for a 30 x 120 matrix.
extern "C" int gpupoly_matrix_transpose(GpuMatrix *out, const GpuMatrix *src) {
    ABI_GUARD_BEGIN
    if (!out || !src) return set_error("gpupoly_matrix_transpose: null matrix");
    if (out == src) return set_error("gpupoly_matrix_transpose: output must not alias the source");
    if (out->ctx != src->ctx || out->level != src->level) return set_error("gpupoly_matrix_transpose: context / level mismatch");
    if (out->rows != src->cols || out->cols != src->rows) return set_error("gpupoly_matrix_transpose: shape mismatch");
    out->format = src->format;
    const size_t polys = matrix_polys(src);
    if (polys == 0) return 0;
    GpuContext *ctx = src->ctx;
    if (ctx_activate(ctx)) return 1;
    const size_t poly_bytes = matrix_limbs(src) * static_cast<size_t>(ctx->N) * ctx->word_bytes;
    if (poly_bytes % 16 != 0 || src->rows == 1 || src->cols == 1) {
        // a vector's transpose is the same bytes; tiny rings (N * word < 16 bytes) go entry by entry
        if (src->rows == 1 || src->cols == 1) {
            HIP_TRY(hipMemcpyAsync(out->data, src->data, src->bytes, hipMemcpyDeviceToDevice, ctx->stream));
            return 0;
        }
        for (size_t r = 0; r < src->rows; ++r)
            for (size_t c = 0; c < src->cols; ++c)
                HIP_TRY(hipMemcpyAsync(static_cast<char *>(out->data) + (c * src->rows + r) * poly_bytes,
                                       static_cast<const char *>(src->data) + (r * src->cols + c) * poly_bytes, poly_bytes,
                                       hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    const size_t vec_per_poly = poly_bytes / 16;
    const unsigned gx = static_cast<unsigned>(std::min<size_t>((vec_per_poly + 255) / 256, 64));
    const size_t gy = std::min<size_t>(polys, 65535), gz = (polys + gy - 1) / gy;
    if (gz > 65535) return set_error("gpupoly_matrix_transpose: matrix too large");
    MXX_LAUNCH(transpose_kernel, dim3(gx, static_cast<unsigned>(gy), static_cast<unsigned>(gz)), dim3(256), 0, ctx->stream,
                       static_cast<uint4 *>(out->data), static_cast<const uint4 *>(src->data), src->rows, src->cols, vec_per_poly);
    HIP_TRY(hipGetLastError());
    return 0;
    ABI_GUARD_END
}

// ---- column blocks <-> one wide matrix in one launch (extensions) ---------------------------------------------------
// The host side of a batch of requests (preimage_batched_sharded: several targets against one trapdoor) concatenates
// the requests' matrices column-wise, makes ONE call, and cuts the result up again.  Through gpu_matrix_copy_block that
// is a launch per request each way - at the small rings of the GGH15 chain the launches ARE the cost.  Here the blocks'
// addresses travel as a kernel argument (64 per launch) and one kernel moves every polynomial, 16 bytes per lane.
#define COLUMN_BLOCKS_MAX 64
struct ColumnBlocks {
    uint32_t count;
    uint32_t start[COLUMN_BLOCKS_MAX + 1];  // first column of block j in the wide matrix; start[count] = one past the last
    void *ptr[COLUMN_BLOCKS_MAX];
};
template <bool SPLIT>
__global__ void column_blocks_kernel(uint4 *__restrict__ whole, ColumnBlocks blocks, size_t rows, size_t cols, size_t vec_per_poly) {
    const size_t span = blocks.start[blocks.count] - blocks.start[0];  // columns this launch covers
    const size_t entry = static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y;
    if (entry >= rows * span) return;
    const size_t r = entry / span;
    const uint32_t c = blocks.start[0] + static_cast<uint32_t>(entry - r * span);
    uint32_t lo = 0, hi = blocks.count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (c >= blocks.start[mid]) lo = mid;
        else hi = mid;
    }
    const size_t bc = blocks.start[lo + 1] - blocks.start[lo];
    uint4 *w = whole + (r * cols + c) * vec_per_poly;
    uint4 *b = static_cast<uint4 *>(blocks.ptr[lo]) + (r * bc + (c - blocks.start[lo])) * vec_per_poly;
    for (size_t v = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < vec_per_poly;
         v += static_cast<size_t>(gridDim.x) * blockDim.x) {
        if (SPLIT) b[v] = w[v];
        else w[v] = b[v];
    }
}

static int column_blocks(GpuMatrix *whole, GpuMatrix *const *blocks, size_t n, bool split, const char *who) {
    if (!whole || (n && !blocks)) return set_error(std::string(who) + ": null argument");
    size_t total = 0;
    for (size_t j = 0; j < n; ++j) {
        const GpuMatrix *b = blocks[j];
        if (!b) return set_error(std::string(who) + ": null block");
        if (b == whole) return set_error(std::string(who) + ": a block must not alias the wide matrix");
        if (b->ctx != whole->ctx || b->level != whole->level) return set_error(std::string(who) + ": context / level mismatch");
        if (b->rows != whole->rows) return set_error(std::string(who) + ": blocks must have the wide matrix's row count");
        if (split ? false : b->format != blocks[0]->format) return set_error(std::string(who) + ": blocks must share one format");
        total += b->cols;
    }
    if (total != whole->cols) return set_error(std::string(who) + ": the blocks' columns must add up to the wide matrix's");
    if (split) {
        for (size_t j = 0; j < n; ++j) blocks[j]->format = whole->format;
    } else if (n) {
        whole->format = blocks[0]->format;
    }
    if (whole->rows == 0 || total == 0) return 0;
    GpuContext *ctx = whole->ctx;
    if (ctx_activate(ctx)) return 1;
    const size_t poly_bytes = matrix_limbs(whole) * static_cast<size_t>(ctx->N) * ctx->word_bytes;
    if (poly_bytes % 16 != 0) {  // rings below 16 bytes per limb vector: block by block through the rectangular copy
        size_t at = 0;
        for (size_t j = 0; j < n; ++j) {
            if (blocks[j]->cols) {
                const int rc = split ? gpu_matrix_copy_block(blocks[j], whole, 0, 0, 0, at, whole->rows, blocks[j]->cols)
                                     : gpu_matrix_copy_block(whole, blocks[j], 0, at, 0, 0, whole->rows, blocks[j]->cols);
                if (rc) return rc;
            }
            at += blocks[j]->cols;
        }
        return 0;
    }
    const size_t vec_per_poly = poly_bytes / 16;
    const unsigned gx = static_cast<unsigned>(std::min<size_t>((vec_per_poly + 255) / 256, 64));
    size_t at = 0, j = 0;
    while (j < n) {
        ColumnBlocks cb;
        cb.count = 0;
        const size_t first = at;
        while (j < n && cb.count < COLUMN_BLOCKS_MAX) {
            if (blocks[j]->cols) {
                cb.start[cb.count] = static_cast<uint32_t>(at);
                cb.ptr[cb.count] = blocks[j]->data;
                ++cb.count;
                at += blocks[j]->cols;
            }
            ++j;
        }
        for (uint32_t t = cb.count; t <= COLUMN_BLOCKS_MAX; ++t) cb.start[t] = static_cast<uint32_t>(at);
        for (uint32_t t = cb.count; t < COLUMN_BLOCKS_MAX; ++t) cb.ptr[t] = nullptr;
        if (cb.count == 0) break;
        const size_t entries = whole->rows * (at - first);
        const size_t gy = std::min<size_t>(entries, 65535), gz = (entries + gy - 1) / gy;
        if (gz > 65535 || at >> 32) return set_error(std::string(who) + ": matrix too large");
        MXX_TRACE_BYTES(2.0 * static_cast<double>(entries) * poly_bytes);
        if (split)
            MXX_LAUNCH(column_blocks_kernel<true>, dim3(gx, static_cast<unsigned>(gy), static_cast<unsigned>(gz)), dim3(256), 0, ctx->stream,
                       static_cast<uint4 *>(whole->data), cb, whole->rows, whole->cols, vec_per_poly);
        else
            MXX_LAUNCH(column_blocks_kernel<false>, dim3(gx, static_cast<unsigned>(gy), static_cast<unsigned>(gz)), dim3(256), 0, ctx->stream,
                       static_cast<uint4 *>(whole->data), cb, whole->rows, whole->cols, vec_per_poly);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// out = [blocks[0] | blocks[1] | ...]: `out` is rows x (sum of the blocks' columns) and takes the blocks' format tag
extern "C" int gpupoly_matrix_concat_columns(GpuMatrix *out, const GpuMatrix *const *blocks, size_t n) {
    ABI_GUARD_BEGIN
    return column_blocks(out, const_cast<GpuMatrix *const *>(blocks), n, false, "gpupoly_matrix_concat_columns");
    ABI_GUARD_END
}

// blocks[j] = the next blocks[j]->cols columns of `src`; the blocks take src's format tag
extern "C" int gpupoly_matrix_split_columns(const GpuMatrix *src, GpuMatrix *const *blocks, size_t n) {
    ABI_GUARD_BEGIN
    return column_blocks(const_cast<GpuMatrix *>(src), blocks, n, true, "gpupoly_matrix_split_columns");
    ABI_GUARD_END
}

// ---- constant matrices written on the device (extensions) ----------------------------------------
// The reference's wrapper builds zero and identity matrices as host byte vectors of the full size (8 bytes per
// residue) and uploads them (src/matrix/gpu_dcrt_poly.rs:343-365 `new_zero_with_state`, :1158-1188 `identity`):
// a PCIe transfer of the whole matrix for a constant.  Here: one memset, plus one small kernel for the diagonal.
extern "C" int gpupoly_matrix_fill_zero(GpuMatrix *out) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpupoly_matrix_fill_zero: null matrix");
    if (out->bytes == 0) return 0;
    if (ctx_activate(out->ctx)) return 1;
    HIP_TRY(hipMemsetAsync(out->data, 0, out->bytes, out->ctx->stream));
    return 0;
    ABI_GUARD_END
}

template <typename W>
__global__ void fill_diagonal_kernel(W *__restrict__ out, const W *__restrict__ scalar, size_t size, size_t words_per_poly) {
    const size_t d = blockIdx.y;
    W *dst = out + (d * size + d) * words_per_poly;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < words_per_poly;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        dst[i] = scalar ? scalar[i] : static_cast<W>(1);  // EVAL form of the constant 1: every slot is 1
}

extern "C" int gpupoly_matrix_fill_identity(GpuMatrix *out, const GpuMatrix *scalar) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpupoly_matrix_fill_identity: null matrix");
    if (out->rows != out->cols) return set_error("gpupoly_matrix_fill_identity: matrix must be square");
    if (scalar) {
        if (scalar->ctx != out->ctx || scalar->level != out->level)
            return set_error("gpupoly_matrix_fill_identity: context/level mismatch");
        if (scalar->rows != 1 || scalar->cols != 1) return set_error("gpupoly_matrix_fill_identity: scalar must be 1x1");
        if (scalar->format != GPU_POLY_FORMAT_EVAL) return set_error("gpupoly_matrix_fill_identity requires an Eval scalar");
        if (scalar == out) return set_error("gpupoly_matrix_fill_identity: output must not alias the scalar");
    }
    out->format = GPU_POLY_FORMAT_EVAL;
    if (out->bytes == 0) return 0;
    GpuContext *ctx = out->ctx;
    if (ctx_activate(ctx)) return 1;
    if (out->rows > 65535) return set_error("gpupoly_matrix_fill_identity: matrix too large");
    HIP_TRY(hipMemsetAsync(out->data, 0, out->bytes, ctx->stream));
    const size_t words = matrix_limbs(out) * static_cast<size_t>(ctx->N);
    const dim3 grid(static_cast<unsigned>(std::min<size_t>((words + 255) / 256, 256)), static_cast<unsigned>(out->rows));
    if (ctx->wide)
        MXX_LAUNCH(fill_diagonal_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream, static_cast<uint64_t *>(out->data),
                           scalar ? static_cast<const uint64_t *>(scalar->data) : nullptr, out->rows, words);
    else
        MXX_LAUNCH(fill_diagonal_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream, static_cast<uint32_t *>(out->data),
                           scalar ? static_cast<const uint32_t *>(scalar->data) : nullptr, out->rows, words);
    HIP_TRY(hipGetLastError());
    return 0;
    ABI_GUARD_END
}

extern "C" int gpupoly_matrix_device_ptr(const GpuMatrix *mat, void **out_ptr, size_t *out_bytes) {
    if (!mat || !out_ptr || !out_bytes) return set_error("gpupoly_matrix_device_ptr: null argument");
    *out_ptr = mat->data;
    *out_bytes = mat->bytes;
    return 0;
}

// ---- device-to-device replica transport (extension) --------------------------
// The reference moves a matrix to another device's context through host bytes
// (to_cpu_staging_bytes -> from_cpu_staging_bytes, src/lookup/ggh15/pubkey_gpu.rs:153-196); its only peer
// copies are the int64 p1 buffer (cuda/src/matrix/MatrixTrapdoor.cu:2816,3422).  Here the replica is one
// peer copy over xGMI (hipMemcpyPeerAsync), ordered on both streams: the copy waits for whatever the source
// context has enqueued on the matrix, and the source context's later work (including a stream-ordered free
// of the source) waits for the copy.
extern "C" int gpupoly_matrix_copy_to_context(GpuContext *dst_ctx, const GpuMatrix *src, GpuMatrix **out) {
    ABI_GUARD_BEGIN
    if (!dst_ctx || !src || !out) return set_error("gpupoly_matrix_copy_to_context: null argument");
    *out = nullptr;
    GpuContext *sctx = src->ctx;
    if (dst_ctx->N != sctx->N || dst_ctx->wide != sctx->wide || src->level >= dst_ctx->limb_count)
        return set_error("gpupoly_matrix_copy_to_context: ring mismatch between the contexts");
    for (int l = 0; l <= src->level; ++l)
        if (dst_ctx->moduli[l] != sctx->moduli[l])
            return set_error("gpupoly_matrix_copy_to_context: modulus mismatch between the contexts");
    GpuMatrix *dst = nullptr;
    if (int rc = gpu_matrix_create(dst_ctx, src->level, src->rows, src->cols, src->format, &dst)) return rc;
    if (src->bytes == 0) {
        *out = dst;
        return 0;
    }
    hipEvent_t ready = nullptr, done = nullptr;
    auto fail = [&](hipError_t e, const char *what) {
        if (ready) (void)hipEventDestroy(ready);
        if (done) (void)hipEventDestroy(done);
        gpu_matrix_destroy(dst);
        return set_error(e, what);
    };
    hipError_t e = hipSetDevice(sctx->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ready, sctx->stream);
    if (e != hipSuccess) return fail(e, "source event");
    e = hipSetDevice(dst_ctx->device);
    if (e != hipSuccess) return fail(e, "hipSetDevice");
    if (dst_ctx->device != sctx->device) {
        int can = 0;
        e = hipDeviceCanAccessPeer(&can, dst_ctx->device, sctx->device);
        if (e != hipSuccess) return fail(e, "hipDeviceCanAccessPeer");
        if (can) {
            e = hipDeviceEnablePeerAccess(sctx->device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) {
                (void)hipGetLastError();
                e = hipSuccess;
            }
            if (e != hipSuccess) return fail(e, "hipDeviceEnablePeerAccess");
        }  // without peer access hipMemcpyPeerAsync stages through the host by itself
    }
    e = hipStreamWaitEvent(dst_ctx->stream, ready, 0);
    if (e == hipSuccess) {
        e = dst_ctx->device == sctx->device
                ? hipMemcpyAsync(dst->data, src->data, src->bytes, hipMemcpyDeviceToDevice, dst_ctx->stream)
                : hipMemcpyPeerAsync(dst->data, dst_ctx->device, src->data, sctx->device, src->bytes, dst_ctx->stream);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(done, dst_ctx->stream);
    if (e != hipSuccess) return fail(e, "peer copy");
    e = hipSetDevice(sctx->device);
    if (e == hipSuccess) e = hipStreamWaitEvent(sctx->stream, done, 0);
    if (e != hipSuccess) return fail(e, "source stream wait");
    (void)hipEventDestroy(ready);  // destruction is deferred by the runtime until the events complete
    (void)hipEventDestroy(done);
    *out = dst;
    return 0;
    ABI_GUARD_END
}

// ---- ABI: RNS batch load/store (MatrixSerde.cu:566-924 behaviour) -----------------------------
static int make_event_set(GpuContext *ctx, GpuEventSet **out_events) {
    GpuEventSet *set = new GpuEventSet();
    set->device = ctx->device;
    set->ctx = ctx;
    set->dev_staging = nullptr;
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        delete set;
        return set_error(e, "hipEventCreate");
    }
    e = hipEventRecord(ev, ctx->stream);
    if (e != hipSuccess) {
        (void)hipEventDestroy(ev);
        delete set;
        return set_error(e, "hipEventRecord");
    }
    set->events.push_back(ev);
    if (out_events) {
        *out_events = set;
    } else {
        (void)hipEventSynchronize(ev);
        (void)hipEventDestroy(ev);
        delete set;
    }
    return 0;
}

extern "C" int gpu_matrix_load_rns_batch(GpuMatrix *mat, const uint8_t *bytes, size_t bytes_per_poly, int format,
                                         GpuEventSet **out_events) {
    ABI_GUARD_BEGIN
    if (out_events) *out_events = nullptr;
    if (!mat) return set_error("gpu_matrix_load_rns_batch: null matrix");
    if (format != GPU_POLY_FORMAT_COEFF && format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpu_matrix_load_rns_batch: invalid format");
    GpuContext *ctx = mat->ctx;
    size_t polys = matrix_polys(mat);
    size_t wpp = matrix_limbs(mat) * static_cast<size_t>(ctx->N);
    if (polys == 0) {
        mat->format = format;
        return 0;
    }
    if (!bytes) return set_error("gpu_matrix_load_rns_batch: null bytes");
    if (bytes_per_poly % 8 != 0 || bytes_per_poly < wpp * 8)
        return set_error("gpu_matrix_load_rns_batch: bytes_per_poly must be a multiple of 8 and >= (level+1)*N*8");
    if (ctx_activate(ctx)) return 1;
    size_t src_wpp = bytes_per_poly / 8;
    // chunked staging keeps the transient footprint bounded for multi-GB matrices
    const size_t max_stage_bytes = size_t(512) << 20;
    size_t polys_per_chunk = std::max<size_t>(1, max_stage_bytes / bytes_per_poly);
    polys_per_chunk = std::min(polys_per_chunk, polys);
    CtxBlock stage_block(ctx);  // back to the cache at scope exit, error paths included
    if (stage_block.alloc(polys_per_chunk * bytes_per_poly)) return 1;
    void *const stage = stage_block.ptr;
    for (size_t p0 = 0; p0 < polys; p0 += polys_per_chunk) {
        size_t pc = std::min(polys_per_chunk, polys - p0);
        HIP_TRY(hipMemcpyAsync(stage, bytes + p0 * bytes_per_poly, pc * bytes_per_poly, hipMemcpyHostToDevice,
                               ctx->stream));
        size_t total = pc * wpp;
        unsigned blocks = grid_for(total, 256);
        if (ctx->wide)
            MXX_LAUNCH(unpack_rns_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                               static_cast<const uint64_t *>(stage), static_cast<uint64_t *>(mat->data) + p0 * wpp,
                               wpp, src_wpp, total);
        else
            MXX_LAUNCH(unpack_rns_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                               static_cast<const uint64_t *>(stage), static_cast<uint32_t *>(mat->data) + p0 * wpp,
                               wpp, src_wpp, total);
        HIP_TRY(hipGetLastError());
    }
    mat->format = format;  // "just retags" 
    return make_event_set(ctx, out_events);  // stage_block returns to the cache here, stream-ordered after its readers
    ABI_GUARD_END
}

extern "C" int gpu_matrix_store_rns_batch(const GpuMatrix *mat, uint8_t *bytes_out, size_t bytes_per_poly, int format,
                                          GpuEventSet **out_events) {
    ABI_GUARD_BEGIN
    if (out_events) *out_events = nullptr;
    if (!mat) return set_error("gpu_matrix_store_rns_batch: null matrix");
    GpuContext *ctx = mat->ctx;
    size_t polys = matrix_polys(mat);
    size_t wpp = matrix_limbs(mat) * static_cast<size_t>(ctx->N);
    if (polys == 0) return 0;
    if (!bytes_out) return set_error("gpu_matrix_store_rns_batch: null output");
    if (format != mat->format)
        return set_error("gpu_matrix_store_rns_batch: format conversion is not supported; convert the matrix first");
    if (bytes_per_poly % 8 != 0 || bytes_per_poly < wpp * 8)
        return set_error("gpu_matrix_store_rns_batch: bytes_per_poly must be a multiple of 8 and >= (level+1)*N*8");
    if (ctx_activate(ctx)) return 1;
    // staged tightly; a padded stride goes out as a 2-D copy that leaves the host's padding bytes untouched, as the
    // reference's cudaMemcpy2DAsync does (MatrixSerde.cu:895-903)
    const size_t tight = wpp * 8;
    const size_t max_stage_bytes = size_t(512) << 20;
    size_t polys_per_chunk = std::max<size_t>(1, max_stage_bytes / tight);
    polys_per_chunk = std::min(polys_per_chunk, polys);
    CtxBlock stage_block(ctx);  // back to the cache at scope exit, error paths included
    if (stage_block.alloc(polys_per_chunk * tight)) return 1;
    void *const stage = stage_block.ptr;
    for (size_t p0 = 0; p0 < polys; p0 += polys_per_chunk) {
        size_t pc = std::min(polys_per_chunk, polys - p0);
        size_t total = pc * wpp;
        unsigned blocks = grid_for(total, 256);
        if (ctx->wide)
            MXX_LAUNCH(pack_rns_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                               static_cast<const uint64_t *>(mat->data) + p0 * wpp, static_cast<uint64_t *>(stage), wpp,
                               wpp, total);
        else
            MXX_LAUNCH(pack_rns_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                               static_cast<const uint32_t *>(mat->data) + p0 * wpp, static_cast<uint64_t *>(stage), wpp,
                               wpp, total);
        HIP_TRY(hipGetLastError());
        if (bytes_per_poly == tight)
            HIP_TRY(hipMemcpyAsync(bytes_out + p0 * bytes_per_poly, stage, pc * tight, hipMemcpyDeviceToHost, ctx->stream));
        else
            HIP_TRY(hipMemcpy2DAsync(bytes_out + p0 * bytes_per_poly, bytes_per_poly, stage, tight, tight, pc,
                                     hipMemcpyDeviceToHost, ctx->stream));
    }
    return make_event_set(ctx, out_events);  // stage_block returns to the cache here, stream-ordered after its readers
    ABI_GUARD_END
}

extern "C" int gpu_matrix_store_const_coeff_batch(const GpuMatrix *mat, uint64_t *words_out, size_t words_per_poly,
                                                  GpuEventSet **out_events) {
    ABI_GUARD_BEGIN
    if (out_events) *out_events = nullptr;
    if (!mat) return set_error("gpu_matrix_store_const_coeff_batch: null matrix");
    GpuContext *ctx = mat->ctx;
    size_t polys = matrix_polys(mat);
    size_t L = matrix_limbs(mat);
    if (polys == 0) return 0;
    if (!words_out) return set_error("gpu_matrix_store_const_coeff_batch: null output");
    if (mat->format != GPU_POLY_FORMAT_COEFF)
        return set_error("gpu_matrix_store_const_coeff_batch requires Coeff format");
    if (words_per_poly < L) return set_error("gpu_matrix_store_const_coeff_batch: words_per_poly < limb count");
    if (ctx_activate(ctx)) return 1;
    // staged as L words per polynomial; a wider stride is a 2-D copy that leaves the words beyond L untouched
    // (MatrixSerde.cu:1041-1049)
    size_t bytes = polys * L * 8;
    CtxBlock stage_block(ctx);
    if (stage_block.alloc(bytes)) return 1;
    void *const stage = stage_block.ptr;
    const dim3 blocks = item_grid(polys * L, 256);
    if (ctx->wide)
        MXX_LAUNCH(const_coeff_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<const uint64_t *>(mat->data), static_cast<uint64_t *>(stage), polys, L,
                           (size_t)ctx->N, L);
    else
        MXX_LAUNCH(const_coeff_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<const uint32_t *>(mat->data), static_cast<uint64_t *>(stage), polys, L,
                           (size_t)ctx->N, L);
    HIP_TRY(hipGetLastError());
    if (words_per_poly == L) HIP_TRY(hipMemcpyAsync(words_out, stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    else HIP_TRY(hipMemcpy2DAsync(words_out, words_per_poly * 8, stage, L * 8, L * 8, polys, hipMemcpyDeviceToHost, ctx->stream));
    return make_event_set(ctx, out_events);  // stage_block returns to the cache here, stream-ordered after its readers
    ABI_GUARD_END
}
