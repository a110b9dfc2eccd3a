// serde.hip — compact wire format: coefficient-domain, CRT-reconstructed, centred integers,
// bit-packed at the matrix-wide maximum width.  Replaces the compact-bytes half of
// cuda/src/matrix/MatrixSerde.cu (ABI: cuda/include/matrix/MatrixSerde.cuh:35-58).
//
// Format (MatrixSerde.cu:280-456,1535-1627; Rust header src/matrix/gpu_dcrt_poly.rs:956-1044):
//   coefficient idx = poly*N + i occupies bits [idx*w, (idx+1)*w) of a little-endian bit
//   stream; its low w-1 bits are |x| and bit w-1 is the sign, where x is the representative
//   of the coefficient in (-Q/2, Q/2] (negative iff value > floor(Q/2)), Q = q_0..q_level;
//   w = max_coeff_bits = 1 + max bit-width of |x| over the matrix (0 for the zero matrix),
//   bytes_per_coeff = ceil(w/8), payload_len = ceil(polys*N*w/8).
//
// Device pipeline: one thread per coefficient does Garner mixed-radix CRT from the limb
// residues (inverse table built at context creation, as Runtime.cu:77-96), Horner-evaluates
// the multi-word integer, centres it, and (pass 1) contributes to the max bit-width or
// (pass 2) ORs its w bits into the zeroed payload.  Nothing is staged per coefficient in HBM:
// the value is recomputed in pass 2 (O(L^2) modmuls, cheaper than a round trip of L words).
#include "common.h"
#include "modarith.h"

#include <algorithm>

static constexpr int kMaxWords = 64;  // up to 4096-bit Q

struct SerdeConsts {
    int limbs;
    int words;                    // 64-bit words per reconstructed coefficient
    uint64_t q[GPUPOLY_MAX_LIMBS];
    uint64_t modulus[kMaxWords];  // Q, little-endian words
    uint64_t half[kMaxWords];     // floor(Q/2)
};

// residues of coefficient (poly, i) -> |x| words (little-endian), sign; returns bit width of |x|
template <typename W>
__device__ __forceinline__ uint32_t reconstruct_centered(const W *__restrict__ src, size_t poly, uint32_t i, uint32_t N,
                                                         const SerdeConsts &sc, const uint64_t *__restrict__ garner,
                                                         size_t garner_stride, uint64_t *x, bool &negative) {
    const int L = sc.limbs, WC = sc.words;
    uint64_t v[GPUPOLY_MAX_LIMBS];
    // Garner: v_k = (r_k - (v_0 + v_1 q_0 + ...)) / (q_0 .. q_{k-1})  mod q_k, computed incrementally
    for (int k = 0; k < L; ++k) {
        const uint64_t qk = sc.q[k];
        uint64_t t = static_cast<uint64_t>(src[(poly * L + k) * N + i]);
        for (int j = 0; j < k; ++j) {
            const uint64_t vj = v[j] % qk;
            const uint64_t d = t >= vj ? t - vj : t + qk - vj;
            t = static_cast<uint64_t>((static_cast<u128_t>(d) * garner[k * garner_stride + j]) % qk);
        }
        v[k] = t;
    }
    // Horner: x = (..(v_{L-1} q_{L-2} + v_{L-2}) q_{L-3} + ..) q_0 + v_0
    for (int w = 0; w < WC; ++w) x[w] = 0;
    x[0] = v[L - 1];
    for (int k = L - 2; k >= 0; --k) {
        const uint64_t m = sc.q[k];
        u128_t carry = v[k];
        for (int w = 0; w < WC; ++w) {
            const u128_t p = static_cast<u128_t>(x[w]) * m + carry;
            x[w] = static_cast<uint64_t>(p);
            carry = p >> 64;
        }
    }
    // centre: negative iff x > floor(Q/2)
    int cmp = 0;
    for (int w = WC - 1; w >= 0; --w) {
        if (x[w] != sc.half[w]) {
            cmp = x[w] > sc.half[w] ? 1 : -1;
            break;
        }
    }
    negative = cmp > 0;
    if (negative) {
        uint64_t borrow = 0;
        for (int w = 0; w < WC; ++w) {
            const uint64_t a = sc.modulus[w], b = x[w];
            const uint64_t d = a - b - borrow;
            borrow = (a < b + borrow) || (b + borrow < b) ? 1 : 0;
            x[w] = d;
        }
    }
    for (int w = WC - 1; w >= 0; --w)
        if (x[w]) return static_cast<uint32_t>(w) * 64u + (64u - static_cast<uint32_t>(__clzll(x[w])));
    return 0;
}

template <typename W>
__global__ void compact_maxbits_kernel(const W *__restrict__ src, size_t polys, uint32_t N, SerdeConsts sc,
                                       const uint64_t *__restrict__ garner, size_t garner_stride,
                                       unsigned int *__restrict__ max_bits) {
    const size_t idx = item_index();
    unsigned int bits = 0;
    if (idx < polys * N) {
        uint64_t x[kMaxWords];
        bool neg;
        bits = reconstruct_centered<W>(src, idx / N, static_cast<uint32_t>(idx % N), N, sc, garner, garner_stride, x, neg);
    }
    // wave-level max, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) bits = max(bits, __shfl_down(bits, off));
    if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_bits, bits);
}

template <typename W>
__global__ void compact_pack_kernel(const W *__restrict__ src, size_t polys, uint32_t N, SerdeConsts sc,
                                    const uint64_t *__restrict__ garner, size_t garner_stride, uint32_t width,
                                    uint32_t *__restrict__ payload_words) {
    const size_t idx = item_index();
    if (idx >= polys * N) return;
    uint64_t x[kMaxWords + 1];
    bool neg;
    reconstruct_centered<W>(src, idx / N, static_cast<uint32_t>(idx % N), N, sc, garner, garner_stride, x, neg);
    // set the sign bit at position width-1
    const uint32_t sb = width - 1;
    for (int w = sc.words; w <= kMaxWords; ++w) x[w] = 0;
    if (neg) x[sb >> 6] |= 1ull << (sb & 63);
    // OR the `width` bits into the stream at bit offset idx*width, 32 bits at a time
    const size_t base = idx * static_cast<size_t>(width);
    uint32_t done = 0;
    while (done < width) {
        const size_t bit = base + done;
        const uint32_t off = static_cast<uint32_t>(bit & 31);
        const uint32_t take = min(32u - off, width - done);
        const uint32_t wi = done >> 6, bo = done & 63;
        uint64_t chunk = x[wi] >> bo;
        if (bo + take > 64) chunk |= x[wi + 1] << (64 - bo);
        const uint32_t val = static_cast<uint32_t>(chunk & ((take == 32) ? 0xffffffffull : ((1ull << take) - 1)));
        if (val) atomicOr(&payload_words[bit >> 5], val << off);
        done += take;
    }
}

template <typename W>
__global__ void compact_unpack_kernel(W *__restrict__ dst, const uint8_t *__restrict__ payload, size_t polys,
                                      uint32_t N, SerdeConsts sc, uint32_t width) {
    const size_t idx = item_index();
    if (idx >= polys * N) return;
    const size_t poly = idx / N;
    const uint32_t i = static_cast<uint32_t>(idx % N);
    const int L = sc.limbs;
    if (width == 0) {
        for (int l = 0; l < L; ++l) dst[(poly * L + l) * N + i] = 0;
        return;
    }
    const size_t base = idx * static_cast<size_t>(width);
    const uint32_t mag_bits = width - 1;
    const size_t sbit = base + mag_bits;
    const bool neg = (payload[sbit >> 3] >> (sbit & 7)) & 1u;
    for (int l = 0; l < L; ++l) {
        const uint64_t q = sc.q[l];
        uint64_t r = 0;
        // most-significant bits first, a byte-aligned chunk at a time
        uint32_t remaining = mag_bits;
        while (remaining) {
            const size_t top = base + remaining;            // one past the highest unread bit
            uint32_t take = static_cast<uint32_t>(top & 7);  // bits of this chunk inside the top byte
            if (take == 0) take = 8;
            take = min(take, remaining);
            const size_t lo = top - take;
            const uint32_t byte = payload[lo >> 3];
            const uint32_t chunk = (byte >> (lo & 7)) & ((1u << take) - 1);
            r = static_cast<uint64_t>(((static_cast<u128_t>(r) << take) + chunk) % q);
            remaining -= take;
        }
        if (neg && r) r = q - r;
        dst[(poly * L + l) * N + i] = static_cast<W>(r);
    }
}

static int build_consts(const GpuMatrix *mat, SerdeConsts &sc) {
    const GpuContext *ctx = mat->ctx;
    const int L = mat->level + 1;
    sc.limbs = L;
    // Q = product of the active moduli, little-endian 64-bit words
    std::vector<uint64_t> Q(1, 1);
    for (int l = 0; l < L; ++l) {
        sc.q[l] = ctx->moduli[l];
        unsigned __int128 carry = 0;
        for (size_t w = 0; w < Q.size(); ++w) {
            unsigned __int128 p = static_cast<unsigned __int128>(Q[w]) * ctx->moduli[l] + carry;
            Q[w] = static_cast<uint64_t>(p);
            carry = p >> 64;
        }
        if (carry) Q.push_back(static_cast<uint64_t>(carry));
    }
    if (Q.size() > static_cast<size_t>(kMaxWords)) return set_error("compact bytes: modulus exceeds 4096 bits");
    sc.words = static_cast<int>(Q.size());
    for (int w = 0; w < kMaxWords; ++w) {
        sc.modulus[w] = w < sc.words ? Q[w] : 0;
    }
    for (int w = 0; w < kMaxWords; ++w) {
        const uint64_t lo = sc.modulus[w] >> 1;
        const uint64_t hi = (w + 1 < kMaxWords) ? (sc.modulus[w + 1] & 1ull) << 63 : 0;
        sc.half[w] = lo | hi;
    }
    return 0;
}

extern "C" int gpu_matrix_store_compact_bytes(GpuMatrix *mat, uint8_t *payload_out, size_t payload_capacity,
                                              uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                              size_t *out_payload_len) {
    ABI_GUARD_BEGIN
    if (!mat || !out_max_coeff_bits || !out_bytes_per_coeff || !out_payload_len)
        return set_error("invalid gpu_matrix_store_compact_bytes arguments");
    *out_max_coeff_bits = 0;
    *out_bytes_per_coeff = 0;
    *out_payload_len = 0;
    GpuContext *ctx = mat->ctx;
    const size_t polys = matrix_polys(mat);
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    // the matrix is converted to COEFF in place (MatrixSerde.cu:1108-1118); the Rust side
    // records the original tag and re-NTTs on load
    if (mat->format == GPU_POLY_FORMAT_EVAL) {
        int rc = gpu_matrix_intt_all(mat);
        if (rc) return rc;
    }
    SerdeConsts sc;
    if (build_consts(mat, sc)) return 1;
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    const size_t coeffs = polys * N;
    const dim3 blocks = item_grid(coeffs, 256);
    const size_t gstride = static_cast<size_t>(ctx->limb_count);
    CtxBlock max_block(ctx);
    if (max_block.alloc(sizeof(unsigned int))) return 1;
    void *const d_max = max_block.ptr;
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(unsigned int), ctx->stream));
    if (ctx->wide)
        hipLaunchKernelGGL(compact_maxbits_kernel<uint64_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<const uint64_t *>(mat->data), polys, N, sc, ctx->d_garner, gstride,
                           static_cast<unsigned int *>(d_max));
    else
        hipLaunchKernelGGL(compact_maxbits_kernel<uint32_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<const uint32_t *>(mat->data), polys, N, sc, ctx->d_garner, gstride,
                           static_cast<unsigned int *>(d_max));
    HIP_TRY(hipGetLastError());
    unsigned int h_max = 0;
    HIP_TRY(hipMemcpyAsync(&h_max, d_max, sizeof(h_max), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const unsigned int width = h_max == 0 ? 0 : h_max + 1;
    if (width > 0xffffu) return set_error("centered max coeff bits exceed u16 range in gpu_matrix_store_compact_bytes");
    const unsigned int bytes_per_coeff = (width + 7) / 8;
    const size_t total_bits = coeffs * static_cast<size_t>(width);
    const size_t payload_len = (total_bits + 7) / 8;
    if (payload_len > payload_capacity) return set_error("payload buffer too small in gpu_matrix_store_compact_bytes");
    if (payload_len > 0) {
        if (!payload_out) return set_error("null payload buffer in gpu_matrix_store_compact_bytes");
        const size_t padded = (payload_len + 3) / 4 * 4 + 4;
        CtxBlock payload_block(ctx);
        if (payload_block.alloc(padded)) return 1;
        void *const d_payload = payload_block.ptr;
        HIP_TRY(hipMemsetAsync(d_payload, 0, padded, ctx->stream));
        if (ctx->wide)
            hipLaunchKernelGGL(compact_pack_kernel<uint64_t>, blocks, dim3(256), 0, ctx->stream,
                               static_cast<const uint64_t *>(mat->data), polys, N, sc, ctx->d_garner, gstride, width,
                               static_cast<uint32_t *>(d_payload));
        else
            hipLaunchKernelGGL(compact_pack_kernel<uint32_t>, blocks, dim3(256), 0, ctx->stream,
                               static_cast<const uint32_t *>(mat->data), polys, N, sc, ctx->d_garner, gstride, width,
                               static_cast<uint32_t *>(d_payload));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(payload_out, d_payload, payload_len, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *out_max_coeff_bits = static_cast<uint16_t>(width);
    *out_bytes_per_coeff = static_cast<uint16_t>(bytes_per_coeff);
    *out_payload_len = payload_len;
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_load_compact_bytes(GpuMatrix *mat, const uint8_t *payload, size_t payload_len,
                                             uint16_t max_coeff_bits) {
    ABI_GUARD_BEGIN
    if (!mat) return set_error("invalid gpu_matrix_load_compact_bytes arguments");
    GpuContext *ctx = mat->ctx;
    const size_t polys = matrix_polys(mat);
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    const size_t coeffs = polys * N;
    if (max_coeff_bits == 0) {
        if (payload_len != 0) return set_error("payload_len must be zero when max_coeff_bits is zero");
    } else if (!payload && coeffs) {
        return set_error("null payload in gpu_matrix_load_compact_bytes");
    }
    const size_t expected = (coeffs * static_cast<size_t>(max_coeff_bits) + 7) / 8;
    if (payload_len != expected) return set_error("payload length mismatch in gpu_matrix_load_compact_bytes");
    mat->format = GPU_POLY_FORMAT_COEFF;
    if (coeffs == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    SerdeConsts sc;
    if (build_consts(mat, sc)) return 1;
    CtxBlock payload_block(ctx);
    if (payload_len) {
        if (payload_block.alloc(payload_len)) return 1;
        HIP_TRY(hipMemcpyAsync(payload_block.ptr, payload, payload_len, hipMemcpyHostToDevice, ctx->stream));
    }
    void *const d_payload = payload_block.ptr;
    const dim3 blocks = item_grid(coeffs, 256);
    if (ctx->wide)
        hipLaunchKernelGGL(compact_unpack_kernel<uint64_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(mat->data), static_cast<const uint8_t *>(d_payload), polys, N, sc,
                           static_cast<uint32_t>(max_coeff_bits));
    else
        hipLaunchKernelGGL(compact_unpack_kernel<uint32_t>, blocks, dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(mat->data), static_cast<const uint8_t *>(d_payload), polys, N, sc,
                           static_cast<uint32_t>(max_coeff_bits));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // synchronous, like the reference; payload may be freed by the caller
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_poly_store_compact_bytes(GpuMatrix *poly, uint8_t *payload_out, size_t payload_capacity,
                                            uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                            size_t *out_payload_len) {
    return gpu_matrix_store_compact_bytes(poly, payload_out, payload_capacity, out_max_coeff_bits, out_bytes_per_coeff,
                                          out_payload_len);
}

extern "C" int gpu_poly_load_compact_bytes(GpuMatrix *poly, const uint8_t *payload, size_t payload_len,
                                           uint16_t max_coeff_bits) {
    return gpu_matrix_load_compact_bytes(poly, payload, payload_len, max_coeff_bits);
}
