// arith.hip — coefficient-wise modular add/sub/mul and matrix x matrix over R_q.
// Replaces cuda/src/matrix/MatrixArith.cu behind cuda/include/matrix/MatrixArith.cuh:10-26.
//
// Matrix product (EVAL domain): for every limb and every evaluation slot i,
//   C[r,c](i) = sum_k A[r,k](i) * B[k,c](i)  mod q_limb
// (reference semantics: src/matrix/base/memory.rs:450-480,589-605 on the CPU side,
//  cuda/src/matrix/MatrixArith.cu:191-287 on the CUDA side).  Slots are the fastest
// axis in HBM, so lanes map to consecutive slots (16 B per lane when SV = 4) and
// each thread keeps a TR x TC register tile of 64-bit lazy accumulators: products
// of two residues < q are summed without reduction for as many terms as fit in
// 64 bits (all of them for 24-bit primes), then reduced once with a mulhi by
// floor(2^64/q).  One launch covers every limb (the reference launches per limb).
#include "common.h"
#include "modarith.h"

#include <algorithm>
#include <vector>

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_NEG = 3 };

template <typename W, int OP, bool BCAST, int VN>
__global__ void elementwise_kernel(W *__restrict__ out, const W *__restrict__ a, const W *__restrict__ b,
                                   const LimbConst *__restrict__ limbs, uint32_t L, uint32_t logN,
                                   size_t words_per_poly, size_t total_vecs) {
    size_t v = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    for (; v < total_vecs; v += stride) {
        const size_t w0 = v * VN;
        const uint32_t limb = static_cast<uint32_t>((w0 >> logN) % L);
        const LimbConst lc = limbs[limb];
        const W q = static_cast<W>(lc.q);
        const size_t bw0 = BCAST ? (w0 % words_per_poly) : w0;
        W av[VN], bv[VN], ov[VN];
        if (VN == 1) {
            av[0] = a[w0];
            bv[0] = b[bw0];
        } else {
            typedef typename std::conditional<sizeof(W) == 4, uint4, ulonglong2>::type V16;
            *reinterpret_cast<V16 *>(av) = *reinterpret_cast<const V16 *>(a + w0);
            *reinterpret_cast<V16 *>(bv) = *reinterpret_cast<const V16 *>(b + bw0);
        }
#pragma unroll
        for (int j = 0; j < VN; ++j) {
            if (OP == OP_ADD) ov[j] = add_mod<W>(av[j], bv[j], q);
            else if (OP == OP_SUB) ov[j] = sub_mod<W>(av[j], bv[j], q);
            else if (OP == OP_NEG) ov[j] = bv[j] ? static_cast<W>(q - bv[j]) : static_cast<W>(0);
            else ov[j] = mul_mod<W>(av[j], bv[j], q, lc.mu, lc.kbits);
        }
        if (VN == 1) {
            out[w0] = ov[0];
        } else {
            typedef typename std::conditional<sizeof(W) == 4, uint4, ulonglong2>::type V16;
            *reinterpret_cast<V16 *>(out + w0) = *reinterpret_cast<const V16 *>(ov);
        }
    }
}

// Kronecker product of two EVAL matrices: out[(i*rb + r), (j*cb + c)] = a[i][j] o b[r][c], one launch
// (blockIdx.y/z = output polynomial, blockIdx.x strides its residues 16 bytes per lane)
template <typename W, int VN>
__global__ void tensor_kernel(W *__restrict__ out, const W *__restrict__ a, const W *__restrict__ b,
                              const LimbConst *__restrict__ limbs, uint32_t logN, size_t words_per_poly, size_t ca,
                              size_t rb, size_t cb, size_t out_polys) {
    const size_t o = static_cast<size_t>(blockIdx.z) * gridDim.y + blockIdx.y;
    if (o >= out_polys) return;
    const size_t out_cols = ca * cb;
    const size_t R = o / out_cols, C = o - R * out_cols;
    const size_t i = R / rb, r = R - i * rb, j = C / cb, c = C - j * cb;
    const W *pa = a + (i * ca + j) * words_per_poly;
    const W *pb = b + (r * cb + c) * words_per_poly;
    W *po = out + o * words_per_poly;
    typedef typename std::conditional<sizeof(W) == 4, uint4, ulonglong2>::type V16;
    for (size_t w0 = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * VN; w0 < words_per_poly;
         w0 += static_cast<size_t>(gridDim.x) * blockDim.x * VN) {
        const LimbConst lc = limbs[w0 >> logN];
        const W q = static_cast<W>(lc.q);
        W av[VN], bv[VN], ov[VN];
        if (VN == 1) {
            av[0] = pa[w0];
            bv[0] = pb[w0];
        } else {
            *reinterpret_cast<V16 *>(av) = *reinterpret_cast<const V16 *>(pa + w0);
            *reinterpret_cast<V16 *>(bv) = *reinterpret_cast<const V16 *>(pb + w0);
        }
#pragma unroll
        for (int t = 0; t < VN; ++t) ov[t] = mul_mod<W>(av[t], bv[t], q, lc.mu, lc.kbits);
        if (VN == 1) po[w0] = ov[0];
        else *reinterpret_cast<V16 *>(po + w0) = *reinterpret_cast<const V16 *>(ov);
    }
}

// one atomic per wave that saw a difference, and nothing more once the flag is up (with every lane of 8192 blocks
// reporting its own difference, comparing two unequal 13 MB matrices took 0.4 ms of serialised atomics)
template <typename W>
__global__ void equal_kernel(const W *__restrict__ a, const W *__restrict__ b, size_t total, int *__restrict__ diff) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    int d = 0;
    for (; i < total; i += stride) {
        d |= (a[i] != b[i]);
        if (__any(d)) break;
    }
    if (__any(d) && (threadIdx.x & 63u) == 0 && *reinterpret_cast<volatile int *>(diff) == 0) atomicOr(diff, 1);
}

// ---- matrix product ---------------------------------------------------------------------
template <typename W, int TR, int TC, int SV, bool PF = false, bool NTB = false>
__global__ void __launch_bounds__(256)
    matmul_kernel(W *__restrict__ C, const W *__restrict__ A, const W *__restrict__ B,
                  const LimbConst *__restrict__ limbs, uint32_t rows, uint32_t inner, uint32_t cols, uint32_t L,
                  uint32_t N, uint32_t col_tiles) {
    const uint32_t limb = blockIdx.z;
    const uint32_t rt = blockIdx.y / col_tiles, ct = blockIdx.y - rt * col_tiles;
    const uint32_t r0 = rt * TR, c0 = ct * TC;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * SV;
    if (i >= N) return;
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q);
    typedef typename std::conditional<sizeof(W) * SV == 16, uint4, typename std::conditional<sizeof(W) * SV == 8, uint2, W>::type>::type VT;
    static_assert(sizeof(VT) == sizeof(W) * SV, "vector width");

    const size_t strideA = static_cast<size_t>(L) * N;  // words between consecutive polys
    // clamp out-of-range tile rows/cols to a valid entry; their results are never stored
    size_t a_off[TR], b_off[TC];
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        uint32_t rr = min(r0 + r, rows - 1);
        a_off[r] = (static_cast<size_t>(rr) * inner * L + limb) * N + i;
    }
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        uint32_t cc = min(c0 + c, cols - 1);
        b_off[c] = (static_cast<size_t>(cc) * L + limb) * N + i;
    }
    const size_t strideBk = static_cast<size_t>(cols) * L * N;

    if constexpr (sizeof(W) == 4) {
        uint64_t acc[TR][TC][SV];
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int s = 0; s < SV; ++s) acc[r][c][s] = 0;
        const uint32_t lazy = lc.lazy_terms;
        uint32_t pending = 0;
        // PF: two operand sets, the loads of term k+1 are issued before term k is multiplied.  The lazy-reduction
        // branch between two iterations keeps the compiler from hoisting them, and a grid with few workgroups - one
        // rank's column block of a sharded product - then has loads in flight only half of the time
        // ((1 x 30)(30 x 15), L = 15: 129 -> 87 us).  Large grids hide that with occupancy and lose 3-6 % to the extra
        // registers, so the host picks PF for small grids only.
        W av[PF ? 2 : 1][TR][SV], bv[PF ? 2 : 1][TC][SV];
        auto load = [&](int set, uint32_t k) {
#pragma unroll
            for (int r = 0; r < TR; ++r)
                *reinterpret_cast<VT *>(av[set][r]) = *reinterpret_cast<const VT *>(A + a_off[r] + k * strideA);
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                // NTB (the host sets it when there is ONE row tile): B is then streamed exactly once, and non-temporal loads
                // keep it from displacing A (read by every column tile) in L2 / the Infinity Cache - M2A 608-629 -> 588 us.
                // With several row tiles the other tiles re-read B from cache and the hint costs 25-55 %
                // ((96 x 64)(64 x 8): 0.48 -> 0.79 ms).  A template flag: a run-time `c ? nt_load : load` is merged into one
                // plain load.
                typedef uint32_t u32xs __attribute__((ext_vector_type(SV)));
                const u32xs *src = reinterpret_cast<const u32xs *>(B + b_off[c] + k * strideBk);
                u32xs t;
                if constexpr (NTB) t = __builtin_nontemporal_load(src);
                else t = *src;
#pragma unroll
                for (int s_ = 0; s_ < SV; ++s_) bv[set][c][s_] = t[s_];
            }
        };
        auto mac = [&](int set) {
#pragma unroll
            for (int r = 0; r < TR; ++r)
#pragma unroll
                for (int c = 0; c < TC; ++c)
#pragma unroll
                    for (int s = 0; s < SV; ++s)
                        acc[r][c][s] += static_cast<uint64_t>(av[set][r][s]) * static_cast<uint64_t>(bv[set][c][s]);
            if (++pending == lazy) {
                pending = 0;
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < TC; ++c)
#pragma unroll
                        for (int s = 0; s < SV; ++s) acc[r][c][s] = reduce_u64_sum(acc[r][c][s], q, lc.mu64);
            }
        };
        if constexpr (PF) {
            load(0, 0);
            for (uint32_t k = 0; k < inner; k += 2) {  // the prefetch past the end re-reads the last term and is dropped
                load(1, min(k + 1, inner - 1));
                mac(0);
                if (k + 1 >= inner) break;
                load(0, min(k + 2, inner - 1));
                mac(1);
            }
        } else {
            for (uint32_t k = 0; k < inner; ++k) {
                load(0, k);
                mac(0);
            }
        }
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (r0 + r >= rows) continue;
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                if (c0 + c >= cols) continue;
                W o[SV];
#pragma unroll
                for (int s = 0; s < SV; ++s) o[s] = reduce_u64_sum(acc[r][c][s], q, lc.mu64);
                *reinterpret_cast<VT *>(C + ((static_cast<size_t>(r0 + r) * cols + (c0 + c)) * L + limb) * N + i) =
                    *reinterpret_cast<const VT *>(o);
            }
        }
    } else {
        // 64-bit words: 128-bit lazy accumulators (a 51-bit prime leaves room for 2^26 products before the one
        // reduction; lc.lazy_terms holds the window) instead of a Barrett reduction per term
        u128_t acc[TR][TC][SV];
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int c = 0; c < TC; ++c)
#pragma unroll
                for (int s = 0; s < SV; ++s) acc[r][c][s] = 0;
        const uint32_t lazy = lc.lazy_terms;
        uint32_t pending = 0;
        // small tiles are latency-bound (one dependent pair of loads per k): the operands of KU iterations are loaded
        // before any of them is multiplied, so KU loads are in flight instead of one (16 products (1 x 76)(76 x 4) at
        // n = 256, L = 12: 0.46 -> 0.22 ms; M4 chain step 0.80 -> 0.73 ms).  The tail past `inner` re-reads the last valid operands and is not accumulated.
        constexpr uint32_t KU = TR * TC * SV <= 8 ? 8 : 1;
        for (uint32_t k0 = 0; k0 < inner; k0 += KU) {
            W av[KU][TR][SV], bv[KU][TC][SV];
#pragma unroll
            for (uint32_t u = 0; u < KU; ++u) {
                const uint32_t k = min(k0 + u, inner - 1);
#pragma unroll
                for (int r = 0; r < TR; ++r)
                    *reinterpret_cast<VT *>(av[u][r]) = *reinterpret_cast<const VT *>(A + a_off[r] + k * strideA);
#pragma unroll
                for (int c = 0; c < TC; ++c) {
                    typedef W wxs __attribute__((ext_vector_type(SV)));
                    const wxs *src = reinterpret_cast<const wxs *>(B + b_off[c] + k * strideBk);
                    wxs t;
                    if constexpr (NTB) t = __builtin_nontemporal_load(src);  // see the 32-bit branch
                    else t = *src;
#pragma unroll
                    for (int s_ = 0; s_ < SV; ++s_) bv[u][c][s_] = t[s_];
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < KU; ++u) {
                if (KU > 1 && k0 + u >= inner) break;
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int c = 0; c < TC; ++c)
#pragma unroll
                        for (int s = 0; s < SV; ++s) acc[r][c][s] += static_cast<u128_t>(av[u][r][s]) * bv[u][c][s];
                if (++pending == lazy) {
                    pending = 0;
#pragma unroll
                    for (int r = 0; r < TR; ++r)
#pragma unroll
                        for (int c = 0; c < TC; ++c)
#pragma unroll
                            for (int s = 0; s < SV; ++s)
                                acc[r][c][s] = reduce_u128_sum(acc[r][c][s], q, lc.mu, lc.kbits, lc.mu64);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (r0 + r >= rows) continue;
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                if (c0 + c >= cols) continue;
                W o[SV];
#pragma unroll
                for (int s = 0; s < SV; ++s) o[s] = reduce_u128_sum(acc[r][c][s], q, lc.mu, lc.kbits, lc.mu64);
                *reinterpret_cast<VT *>(C + ((static_cast<size_t>(r0 + r) * cols + (c0 + c)) * L + limb) * N + i) =
                    *reinterpret_cast<const VT *>(o);
            }
        }
    }
}

// ---- LDS-tiled product for fat shapes (u32 words) -------------------------------------------
// Workgroup = 64 consecutive evaluation slots (one per lane) x a 16x16 tile of C; its four
// waves own 8x8 sub-tiles.  Per K-chunk of 4 the workgroup stages A[16 rows][4 k] and
// B[4 k][16 cols] for its 64 slots in LDS (layout [entry][slot][k]: one 16-byte read per
// entry gives a lane its 4 k-values, conflict-free), so every operand word is fetched from
// L2/HBM once per workgroup instead of once per 8x8 register tile.  Double-buffered: the
// next chunk's global loads are in flight while the current one is multiplied; one barrier
// per chunk.  Blocks are numbered so that the tiles of one (limb, slot chunk) — which share
// all their A and B panels — run on the same XCD's L2.
static constexpr int kMmKC = 4;

__global__ void __launch_bounds__(256, 2)
    matmul_lds_kernel_u32(uint32_t *__restrict__ C, const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
                          const LimbConst *__restrict__ limbs, uint32_t rows, uint32_t inner, uint32_t cols,
                          uint32_t L, uint32_t N, uint32_t row_tiles, uint32_t col_tiles, uint32_t slot_chunks,
                          uint32_t xcd_remap) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][2][16][64][kMmKC];  // [buf][A|B][entry][slot][k]
    const uint32_t tiles = row_tiles * col_tiles;
    uint32_t id = blockIdx.x, tile, group;
    if (xcd_remap) {
        const uint32_t xcd = id & 7u, j = id >> 3;
        tile = j % tiles;
        group = (j / tiles) * 8u + xcd;
    } else {
        tile = id % tiles;
        group = id / tiles;
    }
    const uint32_t limb = group / slot_chunks, chunk = group - limb * slot_chunks;
    const uint32_t rt = tile / col_tiles, ct = tile - rt * col_tiles;
    const uint32_t r0 = rt * 16, c0 = ct * 16;
    // the wave index is made an SGPR so that every operand address below is scalar base + lane
    // offset (address arithmetic on the scalar unit, global_load saddr form)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wr = wave >> 1, wc = wave & 1u;
    const uint32_t slot = chunk * 64u + lane;
    const LimbConst lc = limbs[limb];
    const uint32_t q = static_cast<uint32_t>(lc.q);
    const size_t polyw = static_cast<size_t>(L) * N;
    const size_t base = static_cast<size_t>(limb) * N + slot;

    // this wave stages A rows r0+4*wave.. and B cols c0+4*wave.. (4 each) for every k of a chunk
    uint32_t ga[4][kMmKC], gb[4][kMmKC];
    // scalar row/column base pointers, advanced by whole chunks: no multiplies in the loop
    const uint32_t *pa[4], *pb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t rr = min(r0 + 4 * wave + e, rows - 1);
        const uint32_t cc = min(c0 + 4 * wave + e, cols - 1);
        pa[e] = A + static_cast<size_t>(rr) * inner * polyw + static_cast<size_t>(limb) * N + chunk * 64u;
        pb[e] = B + static_cast<size_t>(cc) * polyw + static_cast<size_t>(limb) * N + chunk * 64u;
    }
    const size_t strideA = polyw, strideB = static_cast<size_t>(cols) * polyw;
    auto fetch = [&](uint32_t k0) {
        if (k0 + kMmKC <= inner) {  // full chunk (wave-uniform branch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int kk = 0; kk < kMmKC; ++kk) {
                    ga[e][kk] = pa[e][kk * strideA + lane];
                    gb[e][kk] = pb[e][kk * strideB + lane];
                }
                pa[e] += kMmKC * strideA;
                pb[e] += kMmKC * strideB;
            }
        } else {  // ragged tail: clamp the address, zero the value
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int kk = 0; kk < kMmKC; ++kk) {
                    const bool live = k0 + kk < inner;
                    const uint32_t av = pa[e][(live ? kk : 0) * strideA + lane];
                    const uint32_t bv = pb[e][(live ? kk : 0) * strideB + lane];
                    ga[e][kk] = live ? av : 0u;
                    gb[e][kk] = live ? bv : 0u;
                }
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<uint4 *>(&lds[buf][0][4 * wave + e][lane][0]) = make_uint4(ga[e][0], ga[e][1], ga[e][2], ga[e][3]);
            *reinterpret_cast<uint4 *>(&lds[buf][1][4 * wave + e][lane][0]) = make_uint4(gb[e][0], gb[e][1], gb[e][2], gb[e][3]);
        }
    };

    uint64_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0;

    const uint32_t nchunks = (inner + kMmKC - 1) / kMmKC;
    const uint32_t lazy = lc.lazy_terms;
    uint32_t pending = 0;
    fetch(0);
    stage(0);
    __syncthreads();
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) fetch((ch + 1) * kMmKC);
        uint4 a4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a4[i] = *reinterpret_cast<const uint4 *>(&lds[buf][0][wr * 8 + i][lane][0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint4 b4 = *reinterpret_cast<const uint4 *>(&lds[buf][1][wc * 8 + j][lane][0]);
            // k in the middle, rows innermost: successive v_mad_u64_u32 hit 8 different accumulators,
            // so the 64-bit multiply-add latency is covered without relying on the second wave
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].x) * b4.x;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].y) * b4.y;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].z) * b4.z;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].w) * b4.w;
        }
        pending += kMmKC;
        if (pending + kMmKC > lazy) {
            pending = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = reduce_u64_sum(acc[i][j], q, lc.mu64);
        }
        if (ch + 1 < nchunks) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = r0 + wr * 8 + i;
        if (r >= rows) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t c = c0 + wc * 8 + j;
            if (c >= cols) continue;
            C[(static_cast<size_t>(r) * cols + c) * polyw + base] = reduce_u64_sum(acc[i][j], q, lc.mu64);
        }
    }
}

static int launch_matmul_lds_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    GpuContext *ctx = out->ctx;
    const uint32_t rows = static_cast<uint32_t>(lhs->rows), inner = static_cast<uint32_t>(lhs->cols),
                   cols = static_cast<uint32_t>(rhs->cols);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out)), N = static_cast<uint32_t>(ctx->N);
    const uint32_t row_tiles = (rows + 15) / 16, col_tiles = (cols + 15) / 16, slot_chunks = N / 64;
    const uint64_t groups = static_cast<uint64_t>(L) * slot_chunks;
    const uint64_t blocks = groups * row_tiles * col_tiles;
    if (blocks > 0x7fffffffull) return set_error("gpu_matrix_mul: matrix too large");
    const uint32_t remap = (groups % 8 == 0) ? 1u : 0u;
    ctx->last_kernel = "matmul_lds_kernel_u32 (64 slots x 16x16 tile, operands staged through registers into LDS)";
    MXX_LAUNCH(matmul_lds_kernel_u32, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ctx->stream,
                       static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(lhs->data),
                       static_cast<const uint32_t *>(rhs->data), ctx->d_limbs, rows, inner, cols, L, N, row_tiles,
                       col_tiles, slot_chunks, remap);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename W, int TR, int TC, int SV, bool PF = false>
static int launch_matmul_cfg(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    GpuContext *ctx = out->ctx;
    const uint32_t rows = static_cast<uint32_t>(lhs->rows), inner = static_cast<uint32_t>(lhs->cols),
                   cols = static_cast<uint32_t>(rhs->cols);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out)), N = static_cast<uint32_t>(ctx->N);
    const uint32_t row_tiles = (rows + TR - 1) / TR, col_tiles = (cols + TC - 1) / TC;
    const uint32_t threads = std::min<uint32_t>(256, std::max<uint32_t>(64, N / SV));
    const uint32_t gx = (N / SV + threads - 1) / threads;
    if (static_cast<uint64_t>(row_tiles) * col_tiles > 65535) return set_error("gpu_matrix_mul: matrix too large");
    dim3 grid(gx, row_tiles * col_tiles, L);
    const bool nt = row_tiles == 1 && rhs->bytes > (size_t(1) << 28);
    static const std::string name[2] = {
        std::string("matmul_kernel<") + (sizeof(W) == 4 ? "u32," : "u64,") + std::to_string(TR) + "," + std::to_string(TC) + "," +
            std::to_string(SV) + (PF ? ",loads-ahead" : "") + "> (register tile rows x cols x slots per lane)",
        std::string("matmul_kernel<") + (sizeof(W) == 4 ? "u32," : "u64,") + std::to_string(TR) + "," + std::to_string(TC) + "," +
            std::to_string(SV) + (PF ? ",loads-ahead" : "") + ",nt> (register tile rows x cols x slots per lane, B streamed once with non-temporal loads)"};
    ctx->last_kernel = name[nt ? 1 : 0].c_str();
    // B is read by this one row tile only and cannot live in the 256 MB Infinity Cache: streamed with non-temporal loads
    // (a smaller B is often re-used from cache by the next product - the hint made repeated products on a 134 MB operand
    // 40 % slower)
    if (nt)
        MXX_LAUNCH((matmul_kernel<W, TR, TC, SV, PF, true>), grid, dim3(threads), 0, ctx->stream,
                           static_cast<W *>(out->data), static_cast<const W *>(lhs->data),
                           static_cast<const W *>(rhs->data), ctx->d_limbs, rows, inner, cols, L, N, col_tiles);
    else
        MXX_LAUNCH((matmul_kernel<W, TR, TC, SV, PF, false>), grid, dim3(threads), 0, ctx->stream,
                           static_cast<W *>(out->data), static_cast<const W *>(lhs->data),
                           static_cast<const W *>(rhs->data), ctx->d_limbs, rows, inner, cols, L, N, col_tiles);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_matmul(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    GpuContext *ctx = out->ctx;
    const size_t rows = lhs->rows, cols = rhs->cols;
    const int N = ctx->N;
    if (ctx->wide) {
        // small rings (BASELINE configs[4]: n = 256): a register tile per thread leaves most of the chip idle -
        // (2x72)*(72x4) at L = 12 is 24 waves of 4x4x2 tiles.  Below ~2 waves per SIMD of tiled work, shrink the
        // tile until the grid covers the chip (every output then re-reads its operands from L2, which is cheap there).
        const uint64_t slots = static_cast<uint64_t>(N) * matrix_limbs(out);
        const uint64_t want = 1024ull * 2 * 64;  // lanes for two waves on every SIMD
        if (N >= 2 && slots / 2 * ((rows + 3) / 4) * ((cols + 3) / 4) >= want) {
            if (rows >= 4) return launch_matmul_cfg<uint64_t, 4, 4, 2>(out, lhs, rhs);
            if (rows >= 2) return launch_matmul_cfg<uint64_t, 2, 4, 2>(out, lhs, rhs);
            return launch_matmul_cfg<uint64_t, 1, 4, 2>(out, lhs, rhs);
        }
        if (slots * ((rows + 1) / 2) * ((cols + 1) / 2) >= want) return launch_matmul_cfg<uint64_t, 2, 2, 1>(out, lhs, rhs);
        return launch_matmul_cfg<uint64_t, 1, 1, 1>(out, lhs, rhs);
    }
    {
        // MXX_HIP_MATMUL_PATH = reg | lds | dma forces a kernel family (tests cover all)
        const char force = ctx->env.matmul_path;
        const bool lds_ok = N >= 64 && (N % 64) == 0;
        // MXX_HIP_MATMUL_PATH = reg | lds | dma | mfma forces a kernel family (tests cover all)
        const bool want_mfma = force ? (force == 'm') : false;
        if (want_mfma) {
            const int rc = launch_matmul_mfma_u32(out, lhs, rhs);
            if (rc >= 0) return rc;
        }
        // Shape rules from a sweep of 200 shapes against every forced family (tools/sweep_mm.py, profiles/r02_notes.md):
        // 32-row tiles (the two streamed kernels) when the last row tile is at least three quarters full; of the two,
        // the 32 slots x 32x32 tile with 16 waves (a third less operand traffic per MAC, four-stage ring) when it
        // wastes no more tile area than the 64-slot 32x16 kernel (64^3: 2.15 against 2.43 ms).
        const size_t inner = lhs->cols;
        const bool narrow = cols < 16 && inner < 128;  // a few short columns: the register tile wins at every row count
        const bool rows32 = !narrow && rows >= 24 && (rows % 32 == 0 || rows % 32 >= 24);
        const bool want_wide = force ? (force == 'w') : (rows32 && cols >= 32 && ((cols + 31) / 32) * 2 <= (cols + 15) / 16);
        if (want_wide) {
            const int rc = launch_matmul_dma32_u32(out, lhs, rhs);
            if (rc >= 0) return rc;
        }
        const bool want_dma = force ? (force == 'd' || force == 'w') : (rows32 && cols >= 16);
        if (want_dma) {
            const int rc = launch_matmul_dma_u32(out, lhs, rhs);
            if (rc >= 0) return rc;
        }
        // 1..3 rows: the register tile (4 slots per lane) streams B once and wins wherever its grid reaches the chip; a
        // small ring with a long inner dimension is a serial chain of `inner` load latencies per wave, and the 64-slot
        // LDS tile (four times the workgroups, k in chunks of 4) is faster while it fits one resident round of
        // 512 workgroups (tools/sweep_rowvec.py: n = 4096, L = 8, (1 x 256)(256 x 16): 130 against 175 us)
        const uint64_t lds_blocks = static_cast<uint64_t>(matrix_limbs(out)) * (N / 64) * ((cols + 15) / 16);
        const bool thin = rows <= 2 && cols >= 8 &&  // 3 rows: the 3 x 4 register tile wins on nearly every such shape
                          lds_blocks <= 512 && (inner >= 128 || (rows >= 2 && inner >= 64));
        const bool want_lds = force ? (force == 'l' || force == 'd' || force == 'w') : ((rows >= 9 && cols >= 8 && !narrow) || thin);
        if (lds_ok && want_lds) return launch_matmul_lds_u32(out, lhs, rhs);
    }
    if (N >= 4) {
        switch (ctx->env.matmul_tile) {  // tuning override (tools/sweep_tiles.py)
#define MXX_TILE(R, C, S)                                                                   \
    case (R * 100 + C * 10 + S) * 2: return launch_matmul_cfg<uint32_t, R, C, S>(out, lhs, rhs); \
    case (R * 100 + C * 10 + S) * 2 + 1: return launch_matmul_cfg<uint32_t, R, C, S, true>(out, lhs, rhs);
            MXX_TILE(1, 8, 4) MXX_TILE(1, 4, 4) MXX_TILE(2, 8, 4) MXX_TILE(2, 4, 4) MXX_TILE(3, 4, 4) MXX_TILE(4, 4, 4)
            MXX_TILE(3, 8, 2) MXX_TILE(4, 8, 1) MXX_TILE(8, 8, 1) MXX_TILE(4, 8, 2)
#undef MXX_TILE
            default: break;
        }
        // 5..8 rows: one 8-row register tile, so that B - the large operand of S * G^-1(B) - is streamed exactly once
        // ((8 x 1024)(1024 x 64): 7.7 ms against 8.4 for the 16 x 16 LDS tile, which leaves half its rows idle)
        if (rows > 4 && rows <= 8) return launch_matmul_cfg<uint32_t, 8, 8, 1>(out, lhs, rhs);
        // "ahead" = the loads-ahead form (PF above) for grids too small to hide a load round trip behind other waves
        if (rows >= 4) {
            const uint64_t waves4 = static_cast<uint64_t>((N + 63) / 64) * ((cols + 7) / 8) * matrix_limbs(out);
            return waves4 <= 8192 ? launch_matmul_cfg<uint32_t, 4, 8, 1, true>(out, lhs, rhs) : launch_matmul_cfg<uint32_t, 4, 8, 1>(out, lhs, rhs);
        }
        // 3 rows: one 3 x 4 tile of 16-byte loads (two 2-row tiles read B twice: (3 x 30)(30 x 120), L = 15, 919 -> 752 us;
        // tools/sweep_tiles.py)
        if (rows == 3) {
            const uint64_t waves3 = static_cast<uint64_t>((N / 4 + 63) / 64) * ((cols + 3) / 4) * matrix_limbs(out);
            return waves3 <= 4096 ? launch_matmul_cfg<uint32_t, 3, 4, 4, true>(out, lhs, rhs) : launch_matmul_cfg<uint32_t, 3, 4, 4>(out, lhs, rhs);
        }
        const uint32_t tc = (rows >= 2 || cols >= 8) ? 8 : 4;
        const uint64_t tile_waves = static_cast<uint64_t>((N / 4 + 63) / 64) * ((rows + 1) / 2) * ((cols + tc - 1) / tc) * matrix_limbs(out);
        const bool ahead = tile_waves <= (rows >= 2 ? 1024u : 8192u);  // the 2-row tile drops to 2 waves per SIMD with the second operand set
        if (rows >= 2) return ahead ? launch_matmul_cfg<uint32_t, 2, 8, 4, true>(out, lhs, rhs) : launch_matmul_cfg<uint32_t, 2, 8, 4>(out, lhs, rhs);
        if (cols >= 8) return ahead ? launch_matmul_cfg<uint32_t, 1, 8, 4, true>(out, lhs, rhs) : launch_matmul_cfg<uint32_t, 1, 8, 4>(out, lhs, rhs);
        return ahead ? launch_matmul_cfg<uint32_t, 1, 4, 4, true>(out, lhs, rhs) : launch_matmul_cfg<uint32_t, 1, 4, 4>(out, lhs, rhs);
    }
    return launch_matmul_cfg<uint32_t, 1, 4, 1>(out, lhs, rhs);
}

// ---- host dispatch for element-wise ops ----------------------------------------------------------
template <typename W, int OP, bool BCAST>
static int launch_elementwise_typed(GpuMatrix *out, const GpuMatrix *a, const GpuMatrix *b) {
    GpuContext *ctx = out->ctx;
    const size_t words = matrix_words(out);
    if (words == 0) return 0;
    const size_t wpp = matrix_limbs(out) * static_cast<size_t>(ctx->N);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out));
    constexpr int VNATIVE = 16 / sizeof(W);
    // out written once; a and b read once each (a broadcast operand is one resident ring element; OP_NEG reads one operand)
    MXX_TRACE_BYTES(static_cast<double>(words) * sizeof(W) * (1 + (OP == OP_NEG || BCAST || a == b ? 1 : 2)));
    if (ctx->N >= VNATIVE) {
        const size_t vecs = words / VNATIVE;
        unsigned blocks = static_cast<unsigned>(std::min<size_t>((vecs + 255) / 256, 16384));
        MXX_LAUNCH((elementwise_kernel<W, OP, BCAST, VNATIVE>), dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<W *>(out->data), static_cast<const W *>(a->data),
                           static_cast<const W *>(b->data), ctx->d_limbs, L, ctx->logN, wpp, vecs);
    } else {
        unsigned blocks = static_cast<unsigned>(std::min<size_t>((words + 255) / 256, 16384));
        MXX_LAUNCH((elementwise_kernel<W, OP, BCAST, 1>), dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<W *>(out->data), static_cast<const W *>(a->data),
                           static_cast<const W *>(b->data), ctx->d_limbs, L, ctx->logN, wpp, words);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int OP, bool BCAST>
static int launch_elementwise(GpuMatrix *out, const GpuMatrix *a, const GpuMatrix *b) {
    if (ctx_activate(out->ctx)) return 1;
    return out->ctx->wide ? launch_elementwise_typed<uint64_t, OP, BCAST>(out, a, b)
                          : launch_elementwise_typed<uint32_t, OP, BCAST>(out, a, b);
}

// ---- ABI ---------------------------------------------------------------------------------------
// a window of whole polynomials of `m`, row-major from polynomial `first_poly`, seen as a rows x cols matrix
static GpuMatrix matrix_view(const GpuMatrix *m, size_t first_poly, size_t rows, size_t cols, size_t poly_bytes) {
    GpuMatrix v;
    v.ctx = m->ctx;
    v.level = m->level;
    v.rows = rows;
    v.cols = cols;
    v.format = m->format;
    v.data = static_cast<char *>(m->data) + first_poly * poly_bytes;
    v.bytes = rows * cols * poly_bytes;
    return v;
}

// Extension: out[dst_row .. dst_row + lhs.rows) = lhs + rhs, written straight into a row block of a taller matrix
// (row blocks are contiguous in the row-major layout).  The preimage's x = [p1 + [R;E] z ; p2 + z] otherwise takes
// a copy_block of p and an add_block of the product per part (src/sampler/trapdoor/gpu.rs:340-369): five passes
// over the largest operands instead of three.
extern "C" int gpupoly_matrix_add_rows(GpuMatrix *out, size_t dst_row, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    ABI_GUARD_BEGIN
    if (!out || !lhs || !rhs) return set_error("gpupoly_matrix_add_rows: null matrix");
    if (matrix_check_same_shape(lhs, rhs, "gpupoly_matrix_add_rows")) return 1;
    if (out->ctx != lhs->ctx || out->level != lhs->level) return set_error("gpupoly_matrix_add_rows: context/level mismatch");
    if (out->cols != lhs->cols || dst_row > out->rows || lhs->rows > out->rows - dst_row)
        return set_error("gpupoly_matrix_add_rows: row block out of range");
    if (lhs->format != rhs->format) return set_error("gpupoly_matrix_add_rows: operands must share a format");
    if (out == lhs || out == rhs) return set_error("gpupoly_matrix_add_rows: output must not alias an input");
    const size_t poly_bytes = matrix_limbs(out) * static_cast<size_t>(out->ctx->N) * out->ctx->word_bytes;
    GpuMatrix view = matrix_view(out, dst_row * out->cols, lhs->rows, lhs->cols, poly_bytes);
    int rc = launch_elementwise<OP_ADD, false>(&view, lhs, rhs);
    if (rc) return rc;
    out->format = lhs->format;  // the whole destination takes the source's tag, as copy_block / add_block do
    return 0;
    ABI_GUARD_END
}

// Extension: out[dst_row .. dst_row + coeff.rows) = NTT(coeff) + addend.  `coeff` holds coefficients, `addend` is EVAL
// of the same shape; the transform, the sum and the placement into a row block of a taller matrix are one pass where a
// fused kernel exists (2^14, 32-bit words: ntt14::fwd_add_kernel).  Elsewhere: with consume_coeff != 0 (the caller
// gives `coeff` up: its contents and format tag are unspecified afterwards) a transform of `coeff` in place and a
// three-operand sum - the passes the unfused sequence takes -, else a copy, a transform in place and an addition, with
// `coeff` untouched.  The preimage's bottom block p2 + z is the caller (mxx_amd/trapdoor.py;
// src/sampler/trapdoor/gpu.rs:340-369 transforms z, then adds): at 2^14 z's EVAL form is never materialised.
extern "C" int gpupoly_matrix_ntt_add_rows(GpuMatrix *out, size_t dst_row, GpuMatrix *coeff, const GpuMatrix *addend,
                                           int consume_coeff) {
    ABI_GUARD_BEGIN
    if (!out || !coeff || !addend) return set_error("gpupoly_matrix_ntt_add_rows: null matrix");
    if (matrix_check_same_shape(coeff, addend, "gpupoly_matrix_ntt_add_rows")) return 1;
    if (out->ctx != coeff->ctx || out->level != coeff->level) return set_error("gpupoly_matrix_ntt_add_rows: context/level mismatch");
    if (out->cols != coeff->cols || dst_row > out->rows || coeff->rows > out->rows - dst_row)
        return set_error("gpupoly_matrix_ntt_add_rows: row block out of range");
    if (coeff->format != GPU_POLY_FORMAT_COEFF || addend->format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpupoly_matrix_ntt_add_rows: expects a COEFF matrix and an EVAL addend");
    if (out == coeff || out == addend) return set_error("gpupoly_matrix_ntt_add_rows: output must not alias an input");
    GpuContext *ctx = out->ctx;
    out->format = GPU_POLY_FORMAT_EVAL;  // the whole destination, as add_rows / copy_block do
    const size_t polys = matrix_polys(coeff);
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    const size_t L = matrix_limbs(out);
    const size_t poly_bytes = L * static_cast<size_t>(ctx->N) * ctx->word_bytes;
    GpuMatrix view = matrix_view(out, dst_row * out->cols, coeff->rows, coeff->cols, poly_bytes);
    view.format = GPU_POLY_FORMAT_EVAL;
    if (!ctx->wide) {
        const int rc = launch_ntt_add_u32(ctx, static_cast<uint32_t *>(view.data), static_cast<const uint32_t *>(coeff->data),
                                          static_cast<const uint32_t *>(addend->data), polys * L, static_cast<uint32_t>(L));
        if (rc >= 0) return rc;
    }
    if (consume_coeff) {
        int rc = launch_ntt(ctx, coeff->data, polys * L, static_cast<int>(L), false);
        if (rc) return rc;
        coeff->format = GPU_POLY_FORMAT_EVAL;
        return launch_elementwise<OP_ADD, false>(&view, coeff, addend);
    }
    MXX_TRACED_COPY("copy (device to device)", ctx->stream, 2.0 * view.bytes,
                    HIP_TRY(hipMemcpyAsync(view.data, coeff->data, view.bytes, hipMemcpyDeviceToDevice, ctx->stream)));
    int rc = launch_ntt(ctx, view.data, polys * L, static_cast<int>(L), false);
    if (rc) return rc;
    return launch_elementwise<OP_ADD, false>(&view, &view, addend);  // in place: every word is read, then written
    ABI_GUARD_END
}

extern "C" int gpu_matrix_add(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    ABI_GUARD_BEGIN
    if (matrix_check_same_shape(out, lhs, "gpu_matrix_add") || matrix_check_same_shape(lhs, rhs, "gpu_matrix_add"))
        return 1;
    int rc = launch_elementwise<OP_ADD, false>(out, lhs, rhs);
    if (rc) return rc;
    // The reference tags the result EVAL unconditionally (MatrixArith.cu:2694), which makes a later
    // gpu_matrix_ntt_all on a COEFF+COEFF sum a silent no-op.  The op is format-agnostic, so the
    // result keeps the operands' format; identical for EVAL operands, the only case that works there.
    out->format = rhs->format;
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_sub(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    ABI_GUARD_BEGIN
    if (matrix_check_same_shape(out, lhs, "gpu_matrix_sub") || matrix_check_same_shape(lhs, rhs, "gpu_matrix_sub"))
        return 1;
    int rc = launch_elementwise<OP_SUB, false>(out, lhs, rhs);
    if (rc) return rc;
    out->format = rhs->format;  // see gpu_matrix_add
    return 0;
    ABI_GUARD_END
}

// Extension: out = -src in one pass.  The reference's wrapper uploads a host vector of zeros, clones it and subtracts
// (src/matrix/gpu_dcrt_poly.rs:1890-1897): a PCIe transfer and five passes for a sign change.  out may be src.
extern "C" int gpupoly_matrix_neg(GpuMatrix *out, const GpuMatrix *src) {
    ABI_GUARD_BEGIN
    if (matrix_check_same_shape(out, src, "gpupoly_matrix_neg")) return 1;
    int rc = launch_elementwise<OP_NEG, false>(out, src, src);
    if (rc) return rc;
    out->format = src->format;
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_mul_scalar(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *scalar) {
    ABI_GUARD_BEGIN
    if (!scalar) return set_error("gpu_matrix_mul_scalar: null scalar");
    if (matrix_check_same_shape(out, lhs, "gpu_matrix_mul_scalar")) return 1;
    if (scalar->ctx != lhs->ctx || scalar->level != lhs->level)
        return set_error("gpu_matrix_mul_scalar: context/level mismatch");
    if (scalar->rows != 1 || scalar->cols != 1) return set_error("gpu_matrix_mul_scalar: scalar must be 1x1");
    if (lhs->format != GPU_POLY_FORMAT_EVAL || scalar->format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpu_matrix_mul_scalar requires Eval format");
    int rc = launch_elementwise<OP_MUL, true>(out, lhs, scalar);
    if (rc) return rc;
    out->format = GPU_POLY_FORMAT_EVAL;
    return 0;
    ABI_GUARD_END
}

template <typename W>
static int launch_tensor_typed(GpuMatrix *out, const GpuMatrix *a, const GpuMatrix *b) {
    GpuContext *ctx = out->ctx;
    constexpr int VNATIVE = 16 / sizeof(W);
    const size_t wpp = matrix_limbs(out) * static_cast<size_t>(ctx->N);
    const size_t polys = matrix_polys(out);
    const size_t gy = std::min<size_t>(polys, 65535), gz = (polys + gy - 1) / gy;
    if (gz > 65535) return set_error("gpupoly_matrix_tensor: matrix too large");
    if (ctx->N >= VNATIVE) {
        const dim3 grid(static_cast<unsigned>(std::min<size_t>((wpp / VNATIVE + 255) / 256, 64)), static_cast<unsigned>(gy), static_cast<unsigned>(gz));
        MXX_LAUNCH((tensor_kernel<W, VNATIVE>), grid, dim3(256), 0, ctx->stream, static_cast<W *>(out->data),
                           static_cast<const W *>(a->data), static_cast<const W *>(b->data), ctx->d_limbs, ctx->logN, wpp,
                           a->cols, b->rows, b->cols, polys);
    } else {
        const dim3 grid(static_cast<unsigned>(std::min<size_t>((wpp + 255) / 256, 64)), static_cast<unsigned>(gy), static_cast<unsigned>(gz));
        MXX_LAUNCH((tensor_kernel<W, 1>), grid, dim3(256), 0, ctx->stream, static_cast<W *>(out->data),
                           static_cast<const W *>(a->data), static_cast<const W *>(b->data), ctx->d_limbs, ctx->logN, wpp,
                           a->cols, b->rows, b->cols, polys);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Extension: a (x) b in one launch.  The reference's wrapper runs, per entry of a, an entry slice, a mul_scalar into
// a temporary and a copy_block (src/matrix/gpu_dcrt_poly.rs:1225-1252): 3 launches and two extra passes over b.
extern "C" int gpupoly_matrix_tensor(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    ABI_GUARD_BEGIN
    if (!out || !lhs || !rhs) return set_error("gpupoly_matrix_tensor: null matrix");
    if (out->ctx != lhs->ctx || out->ctx != rhs->ctx) return set_error("gpupoly_matrix_tensor: context mismatch");
    if (out->level != lhs->level || out->level != rhs->level) return set_error("gpupoly_matrix_tensor: level mismatch");
    if (out->rows != lhs->rows * rhs->rows || out->cols != lhs->cols * rhs->cols)
        return set_error("gpupoly_matrix_tensor: shape mismatch");
    if (lhs->format != GPU_POLY_FORMAT_EVAL || rhs->format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpupoly_matrix_tensor requires Eval format");
    if (out == lhs || out == rhs) return set_error("gpupoly_matrix_tensor: output must not alias an input");
    out->format = GPU_POLY_FORMAT_EVAL;
    if (matrix_polys(out) == 0) return 0;
    if (ctx_activate(out->ctx)) return 1;
    return out->ctx->wide ? launch_tensor_typed<uint64_t>(out, lhs, rhs) : launch_tensor_typed<uint32_t>(out, lhs, rhs);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_mul(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    ABI_GUARD_BEGIN
    if (!out || !lhs || !rhs) return set_error("gpu_matrix_mul: null matrix");
    if (out->ctx != lhs->ctx || out->ctx != rhs->ctx) return set_error("gpu_matrix_mul: context mismatch");
    if (out->level != lhs->level || out->level != rhs->level) return set_error("gpu_matrix_mul: level mismatch");
    if (lhs->cols != rhs->rows || out->rows != lhs->rows || out->cols != rhs->cols)
        return set_error("gpu_matrix_mul: shape mismatch");
    if (lhs->format != GPU_POLY_FORMAT_EVAL || rhs->format != GPU_POLY_FORMAT_EVAL)
        return set_error("gpu_matrix_mul requires Eval format");
    if (out == lhs || out == rhs) return set_error("gpu_matrix_mul: output must not alias an input");
    out->format = GPU_POLY_FORMAT_EVAL;
    if (matrix_polys(out) == 0) return 0;
    if (ctx_activate(out->ctx)) return 1;
    if (lhs->cols == 0) {
        HIP_TRY(hipMemsetAsync(out->data, 0, out->bytes, out->ctx->stream));
        return 0;
    }
    MXX_TRACE_BYTES(static_cast<double>(lhs->bytes) + rhs->bytes + out->bytes);  // (r m + m c + r c) n L w
    return launch_matmul(out, lhs, rhs);
    ABI_GUARD_END
}

// out = lhs * (I_identity (x) X) with X = rhs (mode 0), G^-1(rhs) (mode 1) or the small G^-1(rhs) (mode 2).
// X is built once per column chunk and used for every identity block: the reference's wrappers recompute the
// decomposition of every column for every block (src/matrix/gpu_dcrt_poly.rs:1392-1412).  Row-vector operands
// (one row: BGG encodings) are multiplied in place through views; taller ones go through one reused slice buffer.
static int mul_tensor_identity_impl(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs, size_t identity_size,
                                    int mode, uint32_t base_bits, const char *who) {
    if (!out || !lhs || !rhs) return set_error(std::string(who) + ": null matrix");
    if (out->ctx != lhs->ctx || out->ctx != rhs->ctx) return set_error(std::string(who) + ": context mismatch");
    if (out->level != lhs->level || out->level != rhs->level) return set_error(std::string(who) + ": level mismatch");
    if (mode != 0 && (base_bits == 0 || base_bits >= 63)) return set_error(std::string(who) + ": invalid base_bits");
    if (identity_size == 0) return set_error(std::string(who) + ": identity_size must be positive");
    if (out == lhs || out == rhs) return set_error(std::string(who) + ": output must not alias an input");
    GpuContext *ctx = out->ctx;
    const size_t L = matrix_limbs(out);
    const size_t dpt = mode == 0 ? 1 : (ctx->crt_bits + base_bits - 1) / base_bits;
    const size_t k = mode == 0 ? 1 : (mode == 1 ? dpt * L : dpt);
    const size_t w = rhs->rows * k;  // columns of lhs per identity block
    if (lhs->cols != w * identity_size || out->rows != lhs->rows || out->cols != rhs->cols * identity_size)
        return set_error(std::string(who) + ": shape mismatch");
    if (lhs->format != GPU_POLY_FORMAT_EVAL || (mode == 0 && rhs->format != GPU_POLY_FORMAT_EVAL))
        return set_error(std::string(who) + " requires Eval format");
    out->format = GPU_POLY_FORMAT_EVAL;
    if (matrix_polys(out) == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    if (w == 0) {
        HIP_TRY(hipMemsetAsync(out->data, 0, out->bytes, ctx->stream));
        return 0;
    }
    const size_t poly_bytes = L * static_cast<size_t>(ctx->N) * ctx->word_bytes;
    size_t chunk = rhs->cols;
    if (mode != 0) {
        // digit-matrix budget: a third of what the device could give us now (cached blocks count as available)
        size_t free_b = 0, total_b = 0, budget = size_t(8) << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::max(budget, (free_b + ctx->cached_bytes) / 3);
        else (void)hipGetLastError();
        chunk = std::min(rhs->cols, std::max<size_t>(1, budget / (w * poly_bytes)));
    }
    const bool row_vector = lhs->rows == 1;
    const bool in_place = row_vector || (identity_size == 1 && chunk == rhs->cols);
    GpuMatrix *lhs_buf = nullptr, *prod = nullptr;
    int rc = 0;
    if (!row_vector && identity_size > 1) rc = gpu_matrix_create(ctx, out->level, lhs->rows, w, GPU_POLY_FORMAT_EVAL, &lhs_buf);
    if (!rc && !in_place) rc = gpu_matrix_create(ctx, out->level, lhs->rows, chunk, GPU_POLY_FORMAT_EVAL, &prod);
    for (size_t c0 = 0; !rc && c0 < rhs->cols; c0 += chunk) {
        const size_t cw = std::min(chunk, rhs->cols - c0);
        const bool whole = cw == rhs->cols;
        GpuMatrix *slice = nullptr, *dec = nullptr;
        if (!whole) {
            rc = gpu_matrix_create(ctx, out->level, rhs->rows, cw, rhs->format, &slice);
            if (!rc) rc = gpu_matrix_copy_block(slice, rhs, 0, 0, 0, c0, rhs->rows, cw);
        }
        const GpuMatrix *x = whole ? rhs : slice;
        if (!rc && mode != 0) {
            rc = gpu_matrix_create(ctx, out->level, w, cw, GPU_POLY_FORMAT_EVAL, &dec);
            if (!rc) rc = mode == 1 ? gpu_matrix_decompose_base(x, base_bits, dec) : gpu_matrix_decompose_base_small(x, base_bits, dec);
            x = dec;
        }
        for (size_t i = 0; !rc && i < identity_size; ++i) {
            GpuMatrix lhs_view = matrix_view(lhs, i * w, 1, w, poly_bytes);
            const GpuMatrix *a = lhs;
            if (row_vector) a = &lhs_view;
            else if (identity_size > 1) {
                rc = gpu_matrix_copy_block(lhs_buf, lhs, 0, 0, 0, i * w, lhs->rows, w);
                a = lhs_buf;
            }
            if (rc) break;
            if (row_vector) {
                GpuMatrix out_view = matrix_view(out, i * rhs->cols + c0, 1, cw, poly_bytes);
                rc = gpu_matrix_mul(&out_view, a, x);
            } else if (in_place) {
                rc = gpu_matrix_mul(out, a, x);
            } else {
                GpuMatrix prod_view = matrix_view(prod, 0, lhs->rows, cw, poly_bytes);
                rc = gpu_matrix_mul(&prod_view, a, x);
                if (!rc) rc = gpu_matrix_copy_block(out, &prod_view, 0, i * rhs->cols + c0, 0, 0, lhs->rows, cw);
            }
        }
        gpu_matrix_destroy(slice);
        gpu_matrix_destroy(dec);
    }
    gpu_matrix_destroy(lhs_buf);
    gpu_matrix_destroy(prod);
    if (rc) return rc;
    out->format = GPU_POLY_FORMAT_EVAL;
    return 0;
}

// S * G^-1(B) in one ABI call (extension; the Rust wrapper loops column chunks of width
// MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH = 1, src/matrix/gpu_dcrt_poly.rs:1414-1493: slice, decompose, product,
// copy_block per chunk, re-reading all of S for every chunk - 80.9 ms for (8x1024) * G^-1(64x64) at n = 2^14, L = 8).
// Here: the digits are generated inside the forward transform's load (ntt14::fwd_digits_kernel; the coefficient-
// domain digit matrix never exists) for ALL columns at once when memory allows - 288 GB of HBM hold the 34 GB digit
// matrix of that shape - so S is read once and the product lands directly in `out` (29.7 ms: 20.4 transforms + 9.2
// product); otherwise as few column chunks as fit.  What is NOT done: accumulating the product in the transform's
// epilogue (8 x 16384 accumulators per workgroup), and overlapping the VALU-bound transforms with the HBM-bound
// product on two streams - built and measured (blocks of source rows, C += S[:, block] D_block on a second,
// higher-priority stream): 30.5-34 ms, never better than back to back, the transform grid fills every CU (4
// workgroups of 36 KB LDS) and the product's workgroups only displace them (profiles/r02_notes.md).
extern "C" int gpupoly_matrix_mul_decompose(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs,
                                            uint32_t base_bits) {
    ABI_GUARD_BEGIN
    return mul_tensor_identity_impl(out, lhs, rhs, 1, 1, base_bits, "gpupoly_matrix_mul_decompose");
    ABI_GUARD_END
}

extern "C" int gpupoly_matrix_mul_decompose_small(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs,
                                                  uint32_t base_bits) {
    ABI_GUARD_BEGIN
    return mul_tensor_identity_impl(out, lhs, rhs, 1, 2, base_bits, "gpupoly_matrix_mul_decompose_small");
    ABI_GUARD_END
}

extern "C" int gpupoly_matrix_mul_tensor_identity(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs,
                                                  size_t identity_size) {
    ABI_GUARD_BEGIN
    return mul_tensor_identity_impl(out, lhs, rhs, identity_size, 0, 0, "gpupoly_matrix_mul_tensor_identity");
    ABI_GUARD_END
}

extern "C" int gpupoly_matrix_mul_tensor_identity_decompose(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs,
                                                            size_t identity_size, uint32_t base_bits) {
    ABI_GUARD_BEGIN
    return mul_tensor_identity_impl(out, lhs, rhs, identity_size, 1, base_bits, "gpupoly_matrix_mul_tensor_identity_decompose");
    ABI_GUARD_END
}

// ---- a level of small products in one launch (extension) -----------------------------------------
// The reference evaluates the gates of a circuit level one ABI call each and gets its parallelism from the device
// count (src/circuit/poly_circuit/eval.rs:269, MXX_CIRCUIT_PARALLEL_GATES = number of GPUs, src/env.rs:31-56).  On the
// small rings its own tests use (n <= 256) every product is a launch-latency-bound kernel of a few waves; here up
// to 64 independent products travel in one launch: the descriptors ride in the kernel-argument segment (no device
// table, no copy), blockIdx.y selects the product, one lane computes one residue of one output entry.
struct MulBatchItem {
    void *c;
    const void *a, *b;
    uint32_t rows, inner, cols, pad;
};
constexpr size_t kMulBatchMax = 64;
struct MulBatchArgs {
    MulBatchItem item[kMulBatchMax];
};

template <typename W>
__global__ void __launch_bounds__(256)
    matmul_batch_kernel(MulBatchArgs args, const LimbConst *__restrict__ limbs, uint32_t L, uint32_t logN) {
    const MulBatchItem it = args.item[blockIdx.y];
    const size_t total = (static_cast<size_t>(it.rows) * it.cols * L) << logN;
    const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const size_t N = static_cast<size_t>(1) << logN;
    const uint32_t i = static_cast<uint32_t>(idx & (N - 1));
    const size_t t = idx >> logN;
    const uint32_t limb = static_cast<uint32_t>(t % L);
    const size_t rc = t / L;
    const uint32_t r = static_cast<uint32_t>(rc / it.cols), c = static_cast<uint32_t>(rc - static_cast<size_t>(r) * it.cols);
    const LimbConst lc = limbs[limb];
    const W q = static_cast<W>(lc.q);
    const size_t poly = static_cast<size_t>(L) << logN;
    const W *pa = static_cast<const W *>(it.a) + static_cast<size_t>(r) * it.inner * poly + (static_cast<size_t>(limb) << logN) + i;
    const W *pb = static_cast<const W *>(it.b) + static_cast<size_t>(c) * poly + (static_cast<size_t>(limb) << logN) + i;
    const size_t stride_b = static_cast<size_t>(it.cols) * poly;
    typename Wide<W>::type acc = 0;
    uint32_t pending = 0;
    constexpr uint32_t KU = 8;  // operands of 8 iterations in flight (one dependent pair of loads per k otherwise)
    for (uint32_t k0 = 0; k0 < it.inner; k0 += KU) {
        W av[KU], bv[KU];
#pragma unroll
        for (uint32_t u = 0; u < KU; ++u) {
            const size_t k = min(k0 + u, it.inner - 1);
            av[u] = pa[k * poly];
            bv[u] = pb[k * stride_b];
        }
#pragma unroll
        for (uint32_t u = 0; u < KU; ++u) {
            if (k0 + u >= it.inner) break;
            acc += static_cast<typename Wide<W>::type>(av[u]) * bv[u];
            if (++pending >= lc.lazy_terms) {
                pending = 0;
                if constexpr (sizeof(W) == 4) acc = reduce_u64_sum(acc, q, lc.mu64);
                else acc = reduce_u128_sum(acc, q, lc.mu, lc.kbits, lc.mu64);
            }
        }
    }
    W out;
    if constexpr (sizeof(W) == 4) out = reduce_u64_sum(acc, q, lc.mu64);
    else out = reduce_u128_sum(acc, q, lc.mu, lc.kbits, lc.mu64);
    static_cast<W *>(it.c)[(static_cast<size_t>(r) * it.cols + c) * poly + (static_cast<size_t>(limb) << logN) + i] = out;
}

extern "C" int gpupoly_matrix_mul_batch(GpuMatrix *const *outs, const GpuMatrix *const *lhss, const GpuMatrix *const *rhss,
                                        size_t count) {
    ABI_GUARD_BEGIN
    if (count == 0) return 0;
    if (!outs || !lhss || !rhss) return set_error("gpupoly_matrix_mul_batch: null array");
    GpuContext *ctx = nullptr;
    int level = 0;
    uint64_t max_work = 0;
    for (size_t p = 0; p < count; ++p) {
        GpuMatrix *out = outs[p];
        const GpuMatrix *lhs = lhss[p], *rhs = rhss[p];
        if (!out || !lhs || !rhs) return set_error("gpupoly_matrix_mul_batch: null matrix");
        if (p == 0) {
            ctx = out->ctx;
            level = out->level;
        }
        if (out->ctx != ctx || lhs->ctx != ctx || rhs->ctx != ctx) return set_error("gpupoly_matrix_mul_batch: context mismatch");
        if (out->level != level || lhs->level != level || rhs->level != level)
            return set_error("gpupoly_matrix_mul_batch: level mismatch");
        if (lhs->cols != rhs->rows || out->rows != lhs->rows || out->cols != rhs->cols)
            return set_error("gpupoly_matrix_mul_batch: shape mismatch");
        if (lhs->format != GPU_POLY_FORMAT_EVAL || rhs->format != GPU_POLY_FORMAT_EVAL)
            return set_error("gpupoly_matrix_mul_batch requires Eval format");
        if (out == lhs || out == rhs) return set_error("gpupoly_matrix_mul_batch: output must not alias an input");
        for (size_t o = 0; o < count; ++o)  // an output another product reads or writes: the products are unordered
            if (o != p && (outs[o] == out || lhss[o] == out || rhss[o] == out))
                return set_error("gpupoly_matrix_mul_batch: an output aliases another product's operand");
        max_work = std::max<uint64_t>(max_work, static_cast<uint64_t>(lhs->rows) * lhs->cols * rhs->cols * matrix_limbs(out) *
                                                    static_cast<uint64_t>(ctx->N));
    }
    if (ctx_activate(ctx)) return 1;
    // large products fill the chip by themselves: one by one through the tuned kernels
    if (max_work > (1ull << 24)) {
        for (size_t p = 0; p < count; ++p) {
            const int rc = gpu_matrix_mul(outs[p], lhss[p], rhss[p]);
            if (rc) return rc;
        }
        return 0;
    }
    const uint32_t L = static_cast<uint32_t>(level + 1);
    for (size_t p0 = 0; p0 < count; p0 += kMulBatchMax) {
        const size_t nb = std::min(kMulBatchMax, count - p0);
        MulBatchArgs args;
        size_t max_total = 0, live = 0;
        for (size_t j = 0; j < nb; ++j) {
            GpuMatrix *out = outs[p0 + j];
            out->format = GPU_POLY_FORMAT_EVAL;
            const size_t total = matrix_polys(out) * L * static_cast<size_t>(ctx->N);
            if (total == 0) continue;
            if (lhss[p0 + j]->cols == 0) {
                HIP_TRY(hipMemsetAsync(out->data, 0, out->bytes, ctx->stream));
                continue;
            }
            MulBatchItem &it = args.item[live++];
            it.c = out->data;
            it.a = lhss[p0 + j]->data;
            it.b = rhss[p0 + j]->data;
            it.rows = static_cast<uint32_t>(out->rows);
            it.inner = static_cast<uint32_t>(lhss[p0 + j]->cols);
            it.cols = static_cast<uint32_t>(out->cols);
            it.pad = 0;
            max_total = std::max(max_total, total);
        }
        if (live == 0) continue;
        const dim3 grid(static_cast<unsigned>((max_total + 255) / 256), static_cast<unsigned>(live));
        if (ctx->wide)
            MXX_LAUNCH(matmul_batch_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream, args, ctx->d_limbs, L, ctx->logN);
        else
            MXX_LAUNCH(matmul_batch_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream, args, ctx->d_limbs, L, ctx->logN);
        HIP_TRY(hipGetLastError());
    }
    return 0;
    ABI_GUARD_END
}

// ---- a level of circuit gates in one call (extension) -------------------------------------------------
// The reference evaluates a circuit level gate by gate (src/circuit/poly_circuit/eval.rs:269-345: Add, Sub, Mul,
// Small/LargeScalarMul each turn into one ABI call per gate, from gate-parallel rayon threads).  On a small ring every
// such call is a launch-latency-bound kernel.  gpupoly_batch takes the independent gates of a level: the products go out
// through gpupoly_matrix_mul_batch (up to 64 per launch), the point-wise gates (add / sub / negate / product by a 1x1
// ring element) up to 64 per launch through the kernel below, decompositions one by one through the tuned path.
struct EwBatchItem {
    void *out;
    const void *a, *b;
    uint64_t words;           // residues of the output
    uint32_t words_per_poly;  // broadcast period of b for MUL_SCALAR
    uint32_t op;              // OP_* ; bit 8: b is one polynomial broadcast over a
};
struct EwBatchArgs {
    EwBatchItem item[kMulBatchMax];
};

template <typename W>
__global__ void __launch_bounds__(256) elementwise_batch_kernel(EwBatchArgs args, const LimbConst *__restrict__ limbs, uint32_t L, uint32_t logN) {
    const EwBatchItem it = args.item[blockIdx.y];
    const uint32_t op = it.op & 0xffu;
    const bool bcast = (it.op & 0x100u) != 0;
    const W *a = static_cast<const W *>(it.a), *b = static_cast<const W *>(it.b);
    W *out = static_cast<W *>(it.out);
    for (size_t w = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; w < it.words; w += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const LimbConst lc = limbs[(w >> logN) % L];
        const W q = static_cast<W>(lc.q);
        const W av = a[w], bv = b[bcast ? w % it.words_per_poly : w];
        W r;
        if (op == OP_ADD) r = add_mod<W>(av, bv, q);
        else if (op == OP_SUB) r = sub_mod<W>(av, bv, q);
        else if (op == OP_NEG) r = bv ? static_cast<W>(q - bv) : static_cast<W>(0);
        else r = mul_mod<W>(av, bv, q, lc.mu, lc.kbits);
        out[w] = r;
    }
}

extern "C" int gpupoly_batch(const GpuBatchOp *ops, size_t count, uint32_t base_bits) {
    ABI_GUARD_BEGIN
    if (count == 0) return 0;
    if (!ops) return set_error("gpupoly_batch: null ops");
    GpuContext *ctx = nullptr;
    for (size_t i = 0; i < count; ++i) {
        const GpuBatchOp &o = ops[i];
        if (!o.out || !o.lhs) return set_error("gpupoly_batch: null matrix");
        if (o.kind < GPUPOLY_OP_MUL || o.kind > GPUPOLY_OP_MUL_DECOMPOSE) return set_error("gpupoly_batch: unknown gate kind");
        const bool unary = o.kind == GPUPOLY_OP_NEG || o.kind == GPUPOLY_OP_DECOMPOSE;
        if (!unary && !o.rhs) return set_error("gpupoly_batch: null right operand");
        if (!ctx) ctx = o.out->ctx;
        if (o.out->ctx != ctx || o.lhs->ctx != ctx || (!unary && o.rhs->ctx != ctx)) return set_error("gpupoly_batch: context mismatch");
        // the gates of a level are unordered: no output may be read or written by another gate
        for (size_t j = 0; j < count; ++j)
            if (j != i && (ops[j].out == o.out || ops[j].lhs == o.out || ops[j].rhs == o.out))
                return set_error("gpupoly_batch: an output aliases another gate's operand");
    }
    if (ctx_activate(ctx)) return 1;
    const uint32_t logN = ctx->logN;
    // products: one batched call over the MUL gates (validated there)
    {
        std::vector<GpuMatrix *> outs;
        std::vector<const GpuMatrix *> ls, rs;
        for (size_t i = 0; i < count; ++i)
            if (ops[i].kind == GPUPOLY_OP_MUL) {
                outs.push_back(ops[i].out);
                ls.push_back(ops[i].lhs);
                rs.push_back(ops[i].rhs);
            }
        if (!outs.empty())
            if (int rc = gpupoly_matrix_mul_batch(outs.data(), ls.data(), rs.data(), outs.size())) return rc;
    }
    // point-wise gates, grouped by level (the kernel takes one limb count per launch)
    std::vector<size_t> ew;
    for (size_t i = 0; i < count; ++i) {
        const GpuBatchOp &o = ops[i];
        if (o.kind == GPUPOLY_OP_ADD || o.kind == GPUPOLY_OP_SUB || o.kind == GPUPOLY_OP_NEG || o.kind == GPUPOLY_OP_MUL_SCALAR) {
            const GpuMatrix *b = o.kind == GPUPOLY_OP_NEG ? o.lhs : o.rhs;
            if (matrix_check_same_shape(o.out, o.lhs, "gpupoly_batch")) return 1;
            if (o.kind == GPUPOLY_OP_MUL_SCALAR) {
                if (b->level != o.lhs->level || b->rows != 1 || b->cols != 1) return set_error("gpupoly_batch: scalar must be 1x1 at the operand's level");
                if (o.lhs->format != GPU_POLY_FORMAT_EVAL || b->format != GPU_POLY_FORMAT_EVAL)
                    return set_error("gpupoly_batch: mul_scalar requires Eval format");
            } else if (matrix_check_same_shape(o.lhs, b, "gpupoly_batch")) {
                return 1;
            }
            if (matrix_words(o.out) > (size_t(1) << 24)) {  // fills the chip by itself: the 16-bytes-per-lane kernels
                int rc = o.kind == GPUPOLY_OP_ADD   ? gpu_matrix_add(o.out, o.lhs, o.rhs)
                         : o.kind == GPUPOLY_OP_SUB ? gpu_matrix_sub(o.out, o.lhs, o.rhs)
                         : o.kind == GPUPOLY_OP_NEG ? gpupoly_matrix_neg(o.out, o.lhs)
                                                    : gpu_matrix_mul_scalar(o.out, o.lhs, o.rhs);
                if (rc) return rc;
                continue;
            }
            ew.push_back(i);
        }
    }
    std::vector<char> done(ew.size(), 0);
    for (size_t first = 0; first < ew.size(); ++first) {
        if (done[first]) continue;
        const int level = ops[ew[first]].out->level;
        EwBatchArgs args;
        size_t live = 0, max_words = 0;
        auto flush = [&]() -> int {
            if (live == 0) return 0;
            const dim3 grid(static_cast<unsigned>(std::min<size_t>((max_words + 255) / 256, 4096)), static_cast<unsigned>(live));
            if (ctx->wide)
                MXX_LAUNCH(elementwise_batch_kernel<uint64_t>, grid, dim3(256), 0, ctx->stream, args, ctx->d_limbs, static_cast<uint32_t>(level + 1), logN);
            else
                MXX_LAUNCH(elementwise_batch_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream, args, ctx->d_limbs, static_cast<uint32_t>(level + 1), logN);
            HIP_TRY(hipGetLastError());
            live = 0;
            max_words = 0;
            return 0;
        };
        for (size_t e = first; e < ew.size(); ++e) {
            const GpuBatchOp &o = ops[ew[e]];
            if (done[e] || o.out->level != level) continue;
            done[e] = 1;
            const GpuMatrix *b = o.kind == GPUPOLY_OP_NEG ? o.lhs : o.rhs;
            // format tags as the single-gate entry points set them
            o.out->format = o.kind == GPUPOLY_OP_MUL_SCALAR ? GPU_POLY_FORMAT_EVAL : b->format;
            const size_t words = matrix_words(o.out);
            if (words == 0) continue;
            EwBatchItem &it = args.item[live++];
            it.out = o.out->data;
            it.a = o.lhs->data;
            it.b = b->data;
            it.words = words;
            it.words_per_poly = static_cast<uint32_t>(matrix_limbs(o.out) << logN);
            it.op = (o.kind == GPUPOLY_OP_ADD ? OP_ADD : o.kind == GPUPOLY_OP_SUB ? OP_SUB : o.kind == GPUPOLY_OP_NEG ? OP_NEG : OP_MUL) |
                    (o.kind == GPUPOLY_OP_MUL_SCALAR ? 0x100u : 0u);
            max_words = std::max(max_words, words);
            if (live == kMulBatchMax)
                if (int rc = flush()) return rc;
        }
        if (int rc = flush()) return rc;
    }
    // decompositions: multi-kernel operations with their own tuned paths
    for (size_t i = 0; i < count; ++i) {
        const GpuBatchOp &o = ops[i];
        if (o.kind == GPUPOLY_OP_DECOMPOSE) {
            if (int rc = gpu_matrix_decompose_base(o.lhs, base_bits, o.out)) return rc;
        } else if (o.kind == GPUPOLY_OP_MUL_DECOMPOSE) {
            if (int rc = gpupoly_matrix_mul_decompose(o.out, o.lhs, o.rhs, base_bits)) return rc;
        }
    }
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_matrix_equal(const GpuMatrix *lhs, const GpuMatrix *rhs, int *out_equal) {
    ABI_GUARD_BEGIN
    if (!out_equal) return set_error("gpu_matrix_equal: null out_equal");
    *out_equal = 0;
    if (!lhs || !rhs) return set_error("gpu_matrix_equal: null matrix");
    // mismatching ctx/level/shape/format is "not equal", not an error (MatrixArith.cu:2909-2980)
    if (lhs->ctx != rhs->ctx || lhs->level != rhs->level || lhs->rows != rhs->rows || lhs->cols != rhs->cols ||
        lhs->format != rhs->format)
        return 0;
    size_t words = matrix_words(lhs);
    if (words == 0 || lhs == rhs) {
        *out_equal = 1;
        return 0;
    }
    GpuContext *ctx = lhs->ctx;
    if (ctx_activate(ctx)) return 1;
    CtxBlock flag_block(ctx);
    if (flag_block.alloc(sizeof(int))) return 1;
    void *const flag = flag_block.ptr;
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    unsigned blocks = static_cast<unsigned>(std::min<size_t>((words + 255) / 256, 8192));
    if (ctx->wide)
        MXX_LAUNCH(equal_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<const uint64_t *>(lhs->data), static_cast<const uint64_t *>(rhs->data), words,
                           static_cast<int *>(flag));
    else
        MXX_LAUNCH(equal_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<const uint32_t *>(lhs->data), static_cast<const uint32_t *>(rhs->data), words,
                           static_cast<int *>(flag));
    HIP_TRY(hipGetLastError());
    int diff = 0;
    HIP_TRY(hipMemcpyAsync(&diff, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out_equal = diff ? 0 : 1;
    return 0;
    ABI_GUARD_END
}
