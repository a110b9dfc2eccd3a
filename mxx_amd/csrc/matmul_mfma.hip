// matmul_mfma.hip — R_q matrix product for fat shapes on the matrix cores (u32 words, 24-bit primes).
//
// north_star: "MFMA used only if a residue-packed matrix-mul tile proves dense enough".  The product
// C[r,c](i) = sum_k A[r,k](i) * B[k,c](i) mod q is, per evaluation slot i, a small dense integer
// product; with every residue centred and split into three balanced byte digits
//     a = d0 + 2^8 d1 + 2^16 d2,  d in [-128, 127]   (bytes of a' + 0x808080, each XOR 0x80)
// it becomes 9 products of int8 matrices accumulating exactly in int32 into 5 significance planes:
// v_mfma_i32_32x32x32_i8, 32 cycles per 32x32x32 block against ~2560 VALU cycles for the same MACs
// with v_mad_u64_u32.  What the matrix cores want, though, is "16 consecutive k of one slot per lane",
// while HBM holds "consecutive slots of one (r, k)": the operands have to be transposed on the way in,
// and on-chip storage bounds how many slots a workgroup can hold (registers: 5 int32 planes per output;
// LDS: 3 bytes per operand element).  The shape that fits a CU:
//   workgroup = 16 slots x a 32 x 32 tile of C, 16 waves, wave w <-> slot w;
//   K in chunks of 32: 1024 threads load 16 slots x (32 + 32) rows x 32 k as 16-byte runs of 4 slots
//   (64-byte runs per polynomial), centre / split / pack the bytes of 8 consecutive k (v_perm_b32) and
//   write them as ds_write_b64 into a per-slot image [plane][row][32 k-bytes] (16-byte halves swizzled
//   by row so that the operand reads are conflict-free); each wave then reads its slot's fragments with
//   6 ds_read_b128 and issues 9 MFMAs; after the last chunk the 5 planes are recombined in 64-bit,
//   Barrett-reduced (about 13 VALU instructions per output), staged through LDS and stored as 64-byte runs.
// Operand traffic: A and B panels are each read by two workgroups (tools/mfma_probe.hip: 64-byte runs
// with one re-reader on the same XCD stream at ~6.4 TB/s delivered); the tile order keeps the four
// tiles that share panels next to each other on one XCD.
// A pipelined second form was built and measured as well (persistent workgroups of 8 waves x 2 slots, K in chunks of
// 16 with v_mfma_i32_32x32x16_i8, two chunks = 131 KB of operand loads in flight across chunk and tile boundaries,
// planes 0/2/4 accumulated straight into three running words inside the matrix core, branch-free loads so that the
// compiler's wait counts leave the younger chunk in flight, epilogue constants and the item list precomputed into
// LDS: 250 VGPRs, no spills, bit-exact): 3.49 ms - the same as this form.  Its phases: loads + conversion alone
// 2.36 ms, everything except the loads 1.33 ms, matrix cores + recombination + stores 0.91 ms; 0.37 ms per limb for
// L = 7, 9, 10 against 0.44 for L = 8 (the 512 KB polynomial stride aliases in L2).  With two chunks always in flight
// the operand stream still delivers only ~3.7-4.1 TB/s (8.6 GB per product), i.e. the transposing access shape - 64-byte
// runs at a polynomial stride, 1024 of them per chunk per CU - is throughput-bound below what the VALU kernel's 256-byte
// rows reach, and deeper prefetch does not change it (profiles/r02_notes.md).  It was removed again; this simpler
// form stays as the measured prototype.
// Limits of this form: rows, cols >= 32 pays; inner <= 128 (the int32 planes and their 32-bit pairwise
// recombination are sized for it); every modulus < 16 711 424 (centred residues must keep |d2| < 128).
// Otherwise launch_matmul falls back to the VALU kernels (matmul_dma.hip, arith.hip).
#include <atomic>
#include <cstdlib>

#include "common.h"
#include "modarith.h"

namespace mmfma {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr uint32_t S = 16;           // slots per workgroup (64-byte runs)
constexpr uint32_t TM = 32, TN = 32; // C tile
constexpr uint32_t KC = 32;          // k per chunk = one MFMA depth
constexpr uint32_t THREADS = 1024;
constexpr uint32_t ROW_BYTES = 32, PLANE_BYTES = (TM + TN) * ROW_BYTES, SLOT_BYTES = 3 * PLANE_BYTES + 32;  // 6176
constexpr uint32_t OPERAND_BYTES = S * SLOT_BYTES;                                                        // 98 816
constexpr uint32_t STAGE_STRIDE = S + 1;  // words per (row, col) entry of the output staging image
constexpr uint32_t STAGE_BYTES = TM * TN * STAGE_STRIDE * 4;                                                // 69 632
constexpr size_t LDS_BYTES = OPERAND_BYTES > STAGE_BYTES ? OPERAND_BYTES : STAGE_BYTES;
constexpr uint32_t MAX_INNER = 128;
constexpr uint64_t MAX_Q = 16711424;  // (q-1)/2 + 0x808080 < 2^24

// centre a residue and return a' + 0x808080 (three balanced digits, biased by 128 each)
__device__ __forceinline__ uint32_t centre_biased(uint32_t a, uint32_t q, uint32_t half) {
    return a + 0x808080u - (a > half ? q : 0u);
}

// bytes (p) of four words -> one word, for p = 0, 1, 2
__device__ __forceinline__ void split4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, uint32_t &p0, uint32_t &p1, uint32_t &p2) {
    // v_perm_b32(hi, lo, sel): byte i of the result = byte sel[i] of {hi:lo} (0..3 = lo, 4..7 = hi)
    const uint32_t x01 = __builtin_amdgcn_perm(t1, t0, 0x05010400u);  // t0.b0 t1.b0 t0.b1 t1.b1
    const uint32_t x23 = __builtin_amdgcn_perm(t3, t2, 0x05010400u);
    const uint32_t y01 = __builtin_amdgcn_perm(t1, t0, 0x07030602u);  // t0.b2 t1.b2 t0.b3 t1.b3
    const uint32_t y23 = __builtin_amdgcn_perm(t3, t2, 0x07030602u);
    p0 = __builtin_amdgcn_perm(x23, x01, 0x05040100u) ^ 0x80808080u;  // x01.b0 x01.b1 x23.b0 x23.b1
    p1 = __builtin_amdgcn_perm(x23, x01, 0x07060302u) ^ 0x80808080u;
    p2 = __builtin_amdgcn_perm(y23, y01, 0x05040100u) ^ 0x80808080u;
}

// MODE (phase timing, tools/time_mfma_phases.py; MODE 0 ships): 1 = no global loads (operands are constants),
// 2 = loads + conversion + LDS image only (no MFMA, no recombination)
template <int MODE>
__global__ void __launch_bounds__(THREADS)
    kernel_u32(uint32_t *__restrict__ C, const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
               const LimbConst *__restrict__ limbs, uint32_t rows, uint32_t inner, uint32_t cols, uint32_t L, uint32_t N,
               uint32_t row_tiles, uint32_t col_tiles, uint32_t slot_chunks, uint32_t xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t tiles = row_tiles * col_tiles;
    uint32_t id = blockIdx.x, tile, group;
    if (xcd_remap) {  // blocks b, b+8, ... share an XCD: the tiles of one (limb, slot chunk) are neighbours there
        const uint32_t xcd = id & 7u, j = id >> 3;
        tile = j % tiles;
        group = (j / tiles) * 8u + xcd;
    } else {
        tile = id % tiles;
        group = id / tiles;
    }
    const uint32_t limb = group / slot_chunks, chunk = group - limb * slot_chunks;
    const uint32_t rt = tile / col_tiles, ct = tile - rt * col_tiles;
    const uint32_t r0 = rt * TM, c0 = ct * TN;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = slot
    const LimbConst lc = limbs[limb];
    const uint32_t q = static_cast<uint32_t>(lc.q), half = (q - 1u) >> 1;
    const size_t polyw = static_cast<size_t>(L) * N;
    const size_t slot_base = static_cast<size_t>(limb) * N + static_cast<size_t>(chunk) * S;

    // ---- loader role: (image row r = A row or 32 + B column, k octet ko, slot quad sq) -------------
    const uint32_t sq = tid & 3u, ko = (tid >> 2) & 3u, r = tid >> 4;
    const bool is_a = r < TM;
    const uint32_t grow = is_a ? r0 + r : c0 + (r - TM);   // global row of A / column of B
    const bool row_ok = is_a ? grow < rows : grow < cols;
    const size_t kstride = is_a ? polyw : static_cast<size_t>(cols) * polyw;
    const uint32_t *src = (is_a ? A + static_cast<size_t>(grow) * inner * polyw : B + static_cast<size_t>(grow) * polyw) +
                          slot_base + sq * 4u;
    // byte offset of this thread's 8-byte unit inside a slot image (plane 0)
    const uint32_t unit_off = r * ROW_BYTES + ((((ko >> 1) ^ ((r >> 3) & 1u)) << 4) | ((ko & 1u) << 3));
    const uint32_t wr_base = sq * 4u * SLOT_BYTES + unit_off;

    // ---- MFMA role: this wave's slot; lane = (row or column l & 31, k half l >> 5) -------------------
    const uint32_t mrow = lane & 31u, mh = lane >> 5;
    const uint32_t a_rd = wave * SLOT_BYTES + mrow * ROW_BYTES + ((mh ^ ((mrow >> 3) & 1u)) << 4);
    const uint32_t b_rd = a_rd + TM * ROW_BYTES;  // rows 32..63 of the image are B's columns; (32 + c) >> 3 & 1 == c >> 3 & 1

    v16i acc[5];
#pragma unroll
    for (int p = 0; p < 5; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0;

    const uint32_t nchunks = (inner + KC - 1) / KC;
    for (uint32_t kc = 0; kc < nchunks; ++kc) {
        // 8 polynomials (consecutive k) x 4 slots per thread
        uint4 raw[8];
        const uint32_t k0 = kc * KC + ko * 8u;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (MODE == 1) raw[kk] = uint4{tid + kk, tid, kk + kc, 7u};
            else if (row_ok && k0 + kk < inner) raw[kk] = *reinterpret_cast<const uint4 *>(src + static_cast<size_t>(k0 + kk) * kstride);
            else raw[kk] = uint4{0u, 0u, 0u, 0u};
        }
        if (kc) __syncthreads();  // every wave is done reading the previous chunk's image
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint32_t w[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const uint32_t a = t == 0 ? raw[kk].x : (t == 1 ? raw[kk].y : (t == 2 ? raw[kk].z : raw[kk].w));
                w[kk] = centre_biased(a, q, half);
            }
            uint32_t lo[3], hi[3];
            split4(w[0], w[1], w[2], w[3], lo[0], lo[1], lo[2]);
            split4(w[4], w[5], w[6], w[7], hi[0], hi[1], hi[2]);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<uint2 *>(lds + wr_base + t * SLOT_BYTES + p * PLANE_BYTES) = uint2{lo[p], hi[p]};
        }
        __syncthreads();
        if (MODE == 2) continue;
        v4i fa[3], fb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            fa[p] = *reinterpret_cast<const v4i *>(lds + a_rd + p * PLANE_BYTES);
            fb[p] = *reinterpret_cast<const v4i *>(lds + b_rd + p * PLANE_BYTES);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i + j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i + j], 0, 0, 0);
    }

    // ---- epilogue: v = sum_p acc[p] 2^(8p), |v| <= Kpad (q/2)^2; v' = v + Bq >= 0 with Bq the next multiple
    // of q; Barrett with sh = bits(q) - 1: qhat = floor((v' >> sh) * floor(2^(32+sh) / q) / 2^32) lies in
    // [Q - 3, Q] (the launcher checks v' < 2^(32+sh)), so v' - qhat q is in [0, 4q)
    const uint32_t sh = lc.kbits - 1u;
    const uint64_t bound = static_cast<uint64_t>(nchunks * KC) * (static_cast<uint64_t>(half) + 1u) * (static_cast<uint64_t>(half) + 1u);
    const uint64_t bq = (bound / q + 1ull) * q;
    const uint32_t mu = static_cast<uint32_t>((1ull << (32u + sh)) / q);
    __syncthreads();  // operand image is dead: reuse the LDS for the output staging image
    uint32_t *stage = reinterpret_cast<uint32_t *>(lds);
    if (MODE == 2) {  // keep the image alive: one word per thread goes out
        C[static_cast<size_t>(blockIdx.x % 1024u) * THREADS + tid] = stage[tid];
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int32_t lo = acc[0][e] + acc[1][e] * 256;   // < 2^31 for inner <= 128: |plane| <= 3 K 2^14
        const int32_t mid = acc[2][e] + acc[3][e] * 256;
        int64_t v = static_cast<int64_t>(mid) * 65536 + static_cast<int64_t>(bq);
        v += lo;
        v += static_cast<int64_t>(acc[4][e]) * 4294967296ll;
        const uint64_t vp = static_cast<uint64_t>(v);
        const uint32_t vt = static_cast<uint32_t>(vp >> sh);
        const uint32_t qhat = __umulhi(vt, mu);
        uint32_t res = static_cast<uint32_t>(vp) - qhat * q;  // [0, 4q)
        res = min(res, res - 2u * q);
        res = min(res, res - q);
        const uint32_t row = (e & 3) + 8 * (e >> 2) + 4 * mh, col = mrow;
        stage[(row * TN + col) * STAGE_STRIDE + wave] = res;
    }
    __syncthreads();
    // 64-byte runs: thread -> (tile entry, slot quad)
    const uint32_t quad = tid & 3u;
#pragma unroll
    for (uint32_t it = 0; it < TM * TN * 4 / THREADS; ++it) {
        const uint32_t ent = (tid >> 2) + it * (THREADS / 4);
        const uint32_t er = ent / TN, ec = ent - er * TN;
        if (r0 + er < rows && c0 + ec < cols) {
            const uint32_t *sp = stage + ent * STAGE_STRIDE + quad * 4u;
            const uint4 v = {sp[0], sp[1], sp[2], sp[3]};
            *reinterpret_cast<uint4 *>(C + (static_cast<size_t>(r0 + er) * cols + (c0 + ec)) * polyw + slot_base + quad * 4u) = v;
        }
    }
}
}  // namespace mmfma

// -1: shape / moduli not supported (the caller falls back to the VALU kernels)
int launch_matmul_mfma_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    GpuContext *ctx = out->ctx;
    const uint32_t rows = static_cast<uint32_t>(lhs->rows), inner = static_cast<uint32_t>(lhs->cols),
                   cols = static_cast<uint32_t>(rhs->cols);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out)), N = static_cast<uint32_t>(ctx->N);
    if (ctx->wide || N < mmfma::S || (N % mmfma::S) != 0 || inner == 0 || inner > mmfma::MAX_INNER) return -1;
    const uint64_t kpad = (inner + mmfma::KC - 1) / mmfma::KC * mmfma::KC;
    for (uint32_t l = 0; l < L; ++l) {
        const uint64_t q = ctx->moduli[l], h = (q - 1) / 2 + 1;
        if (q >= mmfma::MAX_Q || q < 256) return -1;
        // the epilogue's v' = v + Bq < 2 Kpad h^2 + 2q must stay below 2^(32 + bits(q) - 1)
        if (2 * kpad * h * h + 2 * q >= (1ull << (31 + ctx->limbs[l].kbits))) return -1;
    }
    const uint32_t row_tiles = (rows + mmfma::TM - 1) / mmfma::TM, col_tiles = (cols + mmfma::TN - 1) / mmfma::TN;
    const uint32_t slot_chunks = N / mmfma::S;
    const uint64_t groups = static_cast<uint64_t>(L) * slot_chunks;
    const uint64_t blocks = groups * row_tiles * col_tiles;
    if (blocks > 0x7fffffffull) return set_error("gpu_matrix_mul: matrix too large");
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(configured.load() & bit)) {
#ifdef GPUPOLY_PHASE_TIMING
        const void *fns[] = {reinterpret_cast<const void *>(mmfma::kernel_u32<0>), reinterpret_cast<const void *>(mmfma::kernel_u32<1>),
                             reinterpret_cast<const void *>(mmfma::kernel_u32<2>)};
#else
        const void *fns[] = {reinterpret_cast<const void *>(mmfma::kernel_u32<0>)};
#endif
        for (const void *fn : fns)
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(mmfma::LDS_BYTES)));
        configured.fetch_or(bit);
    }
    const uint32_t remap = (groups % 8 == 0) ? 1u : 0u;
    ctx->last_kernel = "mmfma::kernel_u32 (16 slots x 32x32 tile, 9 v_mfma_i32_32x32x32_i8 per 32-deep chunk on balanced int8 digits)";
#define MMFMA_LAUNCH(M)                                                                                                   \
    MXX_LAUNCH(mmfma::kernel_u32<M>, dim3(static_cast<unsigned>(blocks)), dim3(mmfma::THREADS), mmfma::LDS_BYTES, \
                       ctx->stream, static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(lhs->data),          \
                       static_cast<const uint32_t *>(rhs->data), ctx->d_limbs, rows, inner, cols, L, N, row_tiles,        \
                       col_tiles, slot_chunks, remap)
#ifdef GPUPOLY_PHASE_TIMING
    // phase-timing builds of the same kernel for tools/time_mfma_phases.py (make PHASE_TIMING=1): modes 1 and 2 skip
    // phases and produce WRONG results, so they do not exist in the shipped library
    static const int mode = [] {
        const char *e = std::getenv("MXX_HIP_MFMA_MODE");
        return e ? std::atoi(e) : 0;
    }();
    if (mode == 1) MMFMA_LAUNCH(1);
    else if (mode == 2 && out->bytes >= size_t(1024) * mmfma::THREADS * 4) MMFMA_LAUNCH(2);  // mode 2 writes 1024 x THREADS words of C
    else MMFMA_LAUNCH(0);
#else
    MMFMA_LAUNCH(0);
#endif
#undef MMFMA_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}
