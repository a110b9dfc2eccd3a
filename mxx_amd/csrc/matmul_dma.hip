#include <atomic>
// matmul_dma.hip — polynomial-matrix product for fat shapes (u32 words), operands streamed
// global -> LDS without passing through registers, three stages deep.
//
// Same mapping as matmul_lds_kernel_u32 (arith.hip): lanes = 64 consecutive evaluation slots,
// a workgroup owns a tile of C for those slots, 64-bit lazy accumulators.  What differs is how
// operands arrive.  There, every wave loads the next K-chunk into 32 registers at the top of an
// iteration and stores it to LDS at the bottom: one chunk (32 KB per workgroup) in flight, issued
// one MAC phase ahead - less than the HBM latency for the 25-30 % of panel reads that are
// compulsory L2 misses, so the VALU sits idle ~45 % of the time (profiles/r01_pmc_sq_m2b.txt).
// Here `global_load_lds_dwordx4` writes 16 bytes per lane straight into LDS (a wave moves four
// 256-byte (entry, k) rows per instruction), no staging registers exist, and the loads for chunk
// c+2 are issued before the multiplies of chunk c: two chunks (96 KB) in flight per workgroup.
//   workgroup = 512 threads = 8 waves (4 x 2), C tile 32 x 16, each wave an 8 x 8 register tile;
//   stage = A[32 rows][4 k][64 slots] + B[16 cols][4 k][64 slots] = 48 KB, 3 stages = 144 KB;
//   one s_barrier per chunk; s_waitcnt vmcnt(n) by hand (the loads of younger chunks stay in flight).
// Measured (M2b, 64^3, L=8): 2.51 ms against 2.57 ms for the register-staged kernel and a ~1.3 ms
// v_mad_u64_u32 floor (tools/mac_rate.hip: 5.0 cycles per MAC in this register pattern).  Phase
// timers (s_memtime) put only ~26 % of a wave's time in the multiply phase: 40 % waits for the
// chunk's rows and 20 % at the barrier for the other waves' rows - 48 KB per chunk per CU through
// the vector-memory path is the limit (5 TB/s L2 -> CU in aggregate), not latency: issuing two
// chunks ahead instead of one changes nothing.  The next step is traffic per MAC (a larger C tile
// per CU), which needs the accumulators out of the VGPR file.
// Requires inner % 4 == 0 (no way to zero-fill a partial chunk without registers); ragged row /
// column tiles clamp their addresses and skip the store.
#include "common.h"
#include "modarith.h"

#include <utility>

namespace mmdma {
constexpr int KC = 4, TROWS = 32, TCOLS = 16, STAGES = 3;
constexpr uint32_t ROW_WORDS = 64;
constexpr uint32_t A_ROWS = TROWS * KC, B_ROWS = TCOLS * KC, STAGE_ROWS = A_ROWS + B_ROWS;
constexpr uint32_t STAGE_WORDS = STAGE_ROWS * ROW_WORDS;
constexpr size_t LDS_BYTES = static_cast<size_t>(STAGES) * STAGE_WORDS * sizeof(uint32_t);
constexpr int LOADS_PER_WAVE = STAGE_ROWS / 4 / 8;  // 6 instructions per wave per stage

// s_waitcnt with only vmcnt constrained (gfx9 encoding: vmcnt [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8])
#define MMDMA_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 0xF) | (((n) >> 4) << 14) | (0x7 << 4) | (0xF << 8))

// LDS reads are issued by hand: the compiler hoists every ds_read of an unrolled chunk to its top
// (and then runs out of registers), which puts all eight waves on the LDS port at once right after
// the barrier.  Two dwords 256*OFF0 and 256*OFF1 bytes past `addr`:
template <int OFF0, int OFF1>
__device__ __forceinline__ uint64_t ds_read2st64(uint32_t addr) {
    uint64_t v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF0), "n"(OFF1));
    return v;
}
// one dword OFF bytes past `addr`
template <int OFF>
__device__ __forceinline__ uint32_t ds_read_at(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// wait until at most N LDS reads issued after them are outstanding; ties the four words to the wait
template <int N>
__device__ __forceinline__ void lds_wait(uint64_t &x, uint64_t &y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int CNT, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(std::make_integer_sequence<int, CNT>{}, static_cast<F &&>(f));
}

__global__ void __launch_bounds__(512, 2)
    kernel_u32(uint32_t *__restrict__ C, const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
               const LimbConst *__restrict__ limbs, uint32_t rows, uint32_t inner, uint32_t cols, uint32_t L, uint32_t N,
               uint32_t row_tiles, uint32_t col_tiles, uint32_t slot_chunks, uint32_t xcd_remap, uint32_t nt_c) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // [STAGES][STAGE_ROWS][64]
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
    const uint32_t r0 = rt * TROWS, c0 = ct * TCOLS;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wr = wave >> 1, wc = wave & 1u;
    const LimbConst lc = limbs[limb];
    const uint32_t q = static_cast<uint32_t>(lc.q);
    const size_t polyw = static_cast<size_t>(L) * N;
    const size_t slot_base = static_cast<size_t>(limb) * N + chunk * 64u;

    // loader: instruction j of this wave moves rows rho0 .. rho0+3 (rho0 = wave*24 + 4j); a lane
    // supplies 16 bytes: row rho0 + lane/16, slots 4*(lane%16) .. +3
    const uint32_t *src[LOADS_PER_WAVE];
    const size_t stride_a = static_cast<size_t>(KC) * polyw, stride_b = static_cast<size_t>(KC) * cols * polyw;
#pragma unroll
    for (int j = 0; j < LOADS_PER_WAVE; ++j) {
        const uint32_t rho = wave * (4 * LOADS_PER_WAVE) + 4 * j + (lane >> 4);
        const uint32_t part = (lane & 15u) * 4u;
        if (rho < A_ROWS) {
            const uint32_t e = rho >> 2, k = rho & 3u;
            const uint32_t rr = min(r0 + e, rows - 1);
            src[j] = A + (static_cast<size_t>(rr) * inner + k) * polyw + slot_base + part;
        } else {
            const uint32_t rb = rho - A_ROWS, c = rb >> 2, k = rb & 3u;
            const uint32_t cc = min(c0 + c, cols - 1);
            src[j] = B + (static_cast<size_t>(k) * cols + cc) * polyw + slot_base + part;
        }
    }
    auto issue = [&](uint32_t stage) {
#pragma unroll
        for (int j = 0; j < LOADS_PER_WAVE; ++j) {
            const uint32_t rho0 = wave * (4 * LOADS_PER_WAVE) + 4 * j;  // wave-uniform
            __builtin_amdgcn_global_load_lds(src[j], lds + stage * STAGE_WORDS + rho0 * ROW_WORDS, 16, 0, 0);
            src[j] += rho0 < A_ROWS ? stride_a : stride_b;
        }
    };

    uint64_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0;

    const uint32_t nch = inner / KC;
    const uint32_t lazy = lc.lazy_terms;
    uint32_t pending = 0;
    // byte addresses inside LDS (the dynamic segment starts at LDS offset 0: no static __shared__ here)
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
    const uint32_t a_addr = lds0 + ((wr * 8u * KC) * ROW_WORDS + lane) * 4u;
    const uint32_t b_addr = lds0 + ((A_ROWS + wc * 8u * KC) * ROW_WORDS + lane) * 4u;
    constexpr uint32_t STAGE_BYTES = STAGE_WORDS * 4u;

    // One iteration = one K-chunk.  a_cur holds this wave's A fragment (8 rows x 4 k, as pairs) of
    // chunk ch; while column j is multiplied, column j+1's B values and row j of chunk ch+1's A
    // fragment are read, so the LDS port works through the whole multiply phase instead of serving
    // all eight waves back to back after the barrier with every SIMD idle.
    auto step = [&](uint32_t ch, uint32_t stage, uint64_t (&a_cur)[8][2], uint64_t (&a_nxt)[8][2]) {
        const uint32_t next_stage = stage + 1 == STAGES ? 0 : stage + 1;
        const bool more = ch + 1 < nch;
        if (more) MMDMA_WAIT_VM(0);               // chunk ch+1 (issued one iteration ago) has landed
        asm volatile("s_barrier" ::: "memory");   // ... for every wave, and everyone is done with chunk ch-1
        if (ch + 2 < nch) issue(next_stage + 1 == STAGES ? 0 : next_stage + 1);
        const uint32_t sb = b_addr + stage * STAGE_BYTES;
        const uint32_t sa = a_addr + (more ? next_stage : stage) * STAGE_BYTES;  // last chunk: harmless re-read
        uint64_t b[2][2];
        b[0][0] = ds_read2st64<0, 1>(sb);
        b[0][1] = ds_read2st64<2, 3>(sb);
        static_for<8>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int cur = j & 1, nxt = cur ^ 1;
            if constexpr (j + 1 < 8) {
                b[nxt][0] = ds_read2st64<4 * (j + 1), 4 * (j + 1) + 1>(sb);
                b[nxt][1] = ds_read2st64<4 * (j + 1) + 2, 4 * (j + 1) + 3>(sb);
            }
            a_nxt[j][0] = ds_read2st64<4 * j, 4 * j + 1>(sa);
            a_nxt[j][1] = ds_read2st64<4 * j + 2, 4 * j + 3>(sa);
            // younger than b[cur]: the previous step's A row (2, none for j = 0) and this step's 2 or 4
            lds_wait<(j == 0 ? 0 : 2) + (j + 1 < 8 ? 4 : 2)>(b[cur][0], b[cur][1]);
            const uint32_t bk[KC] = {static_cast<uint32_t>(b[cur][0]), static_cast<uint32_t>(b[cur][0] >> 32),
                                     static_cast<uint32_t>(b[cur][1]), static_cast<uint32_t>(b[cur][1] >> 32)};
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t av = (k & 1) ? static_cast<uint32_t>(a_cur[i][k >> 1] >> 32) : static_cast<uint32_t>(a_cur[i][k >> 1]);
                    acc[i][j] += static_cast<uint64_t>(av) * bk[k];
                }
        });
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_wait<0>(a_nxt[i][0], a_nxt[i][1]);  // long since complete
        pending += KC;
        if (pending + KC > lazy) {
            pending = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = reduce_u64_sum(acc[i][j], q, lc.mu64);
        }
    };

    uint64_t a0[8][2], a1[8][2];
    issue(0);
    if (nch > 1) {
        issue(1);
        MMDMA_WAIT_VM(LOADS_PER_WAVE);
    } else {
        MMDMA_WAIT_VM(0);
    }
    asm volatile("s_barrier" ::: "memory");
    static_for<8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        a0[i][0] = ds_read2st64<4 * i, 4 * i + 1>(a_addr);
        a0[i][1] = ds_read2st64<4 * i + 2, 4 * i + 3>(a_addr);
    });
#pragma unroll
    for (int i = 0; i < 8; ++i) lds_wait<0>(a0[i][0], a0[i][1]);
    uint32_t stage = 0;
    for (uint32_t ch = 0; ch < nch; ch += 2) {
        step(ch, stage, a0, a1);
        stage = stage + 1 == STAGES ? 0 : stage + 1;
        if (ch + 1 < nch) {
            step(ch + 1, stage, a1, a0);
            stage = stage + 1 == STAGES ? 0 : stage + 1;
        }
    }
    const BoundedReduce br = bounded_reduce_setup(lc.kbits, lc.mu64, inner);
    // the output pointer is carried along the tile by two strides (see kernel_u32 of mmdma32 below)
    const uint32_t c_first = c0 + wc * 8u;
    uint32_t *const dst00 = C + (static_cast<size_t>(r0 + wr * 8u) * cols + c_first) * polyw + slot_base + lane;
    const size_t row_step = static_cast<size_t>(cols) * polyw;
    auto store_tile = [&](auto reduce) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (r0 + wr * 8 + i >= rows) continue;
            uint32_t *dst = dst00 + i * row_step;
#pragma unroll
            for (int j = 0; j < 8; ++j, dst += polyw) {
                if (c_first + j >= cols) continue;
                if (nt_c) __builtin_nontemporal_store(reduce(acc[i][j]), dst);  // see kernel_u32 of mmdma32 below
                else *dst = reduce(acc[i][j]);
            }
        }
    };
    if (br.ok) store_tile([&](uint64_t v) { return reduce_u64_bounded(v, q, br); });
    else store_tile([&](uint64_t v) { return reduce_u64_sum(v, q, lc.mu64); });
}
}  // namespace mmdma

// ---- the same product with HALF the slots, TWICE the tile and 16 waves ----------------------------------------------
// Operand delivery bounds kernel_u32 above: a workgroup moves (TROWS + TCOLS) x KC rows per TROWS x TCOLS x KC MACs per
// slot, i.e. 4 (32 + 16) / (32 x 16) = 0.375 bytes per MAC = 12.9 GB for M2b through an L2 -> LDS path that delivers
// ~5 TB/s (2.43 ms).  Bytes per MAC depend on the tile's shape only, and the tile is bounded by the accumulators: the
// way to a larger tile is FEWER SLOTS per workgroup.  Here a workgroup owns 32 consecutive slots x a 32 x 32 tile:
// 0.25 bytes per MAC (8.6 GB), global rows of 128 bytes, and the freed LDS holds a four-stage ring (loads issued three
// chunks ahead).  A wave is two halves of 32 lanes on the same 8 tile rows (their A reads coincide: LDS broadcast)
// and on adjacent groups of 4 columns, whose B rows are stored interleaved so that one ds_read covers 256 contiguous
// bytes.  A lane owns 8 rows x 4 columns (64 accumulator registers): 1024 threads, four waves per SIMD - with 8 x 8
// per lane (128 registers, two waves per SIMD, LDS reads software-pipelined by hand) the same tile ran 2.25 ms, this
// form 2.15 ms; no software pipelining of the LDS reads here (no registers left for it), the other waves cover them.
//   stage = A[32 rows][4 k][32 slots] + B[32 columns][4 k][32 slots] = 32 KB, 4 stages = 128 KB
namespace mmdma32 {
using mmdma::ds_read2st64;
using mmdma::lds_wait;
using mmdma::static_for;
constexpr int KC = 4, TROWS = 32, TCOLS = 32, STAGES = 4;
constexpr uint32_t SLOTS = 32;
constexpr uint32_t A_WORDS = TROWS * KC * SLOTS, B_WORDS = TCOLS * KC * SLOTS, STAGE_WORDS = A_WORDS + B_WORDS;
constexpr size_t LDS_BYTES = static_cast<size_t>(STAGES) * STAGE_WORDS * sizeof(uint32_t);
constexpr int LOADS_PER_WAVE = (TROWS + TCOLS) * KC / 8 / 16;  // 2 instructions (8 rows of 128 bytes each) per wave per stage

__global__ void __launch_bounds__(1024, 4)
    kernel_u32(uint32_t *__restrict__ C, const uint32_t *__restrict__ A, const uint32_t *__restrict__ B,
               const LimbConst *__restrict__ limbs, uint32_t rows, uint32_t inner, uint32_t cols, uint32_t L, uint32_t N,
               uint32_t row_tiles, uint32_t col_tiles, uint32_t slot_chunks, uint32_t xcd_remap, uint32_t nt_c) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
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
    const uint32_t r0 = rt * TROWS, c0 = ct * TCOLS;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t slot = lane & 31u, half = lane >> 5;
    const uint32_t wr = wave >> 2, wq = wave & 3u;  // tile rows 8 wr .., columns 8 wq + 4 half ..
    const LimbConst lc = limbs[limb];
    const uint32_t q = static_cast<uint32_t>(lc.q);
    const size_t polyw = static_cast<size_t>(L) * N;
    const size_t slot_base = static_cast<size_t>(limb) * N + chunk * SLOTS;

    // LDS rows of 128 bytes.  0..127: A, row = i * 4 + k.  128..255: B, row - 128 = ((cq * 4 + cj) * 4 + k) * 2 + ch for
    // tile column 8 cq + 4 ch + cj: the two half-waves of a wave read adjacent rows.
    const uint32_t *src[LOADS_PER_WAVE];
    const size_t stride_a = static_cast<size_t>(KC) * polyw, stride_b = static_cast<size_t>(KC) * cols * polyw;
#pragma unroll
    for (int j = 0; j < LOADS_PER_WAVE; ++j) {
        const uint32_t rho = (wave * LOADS_PER_WAVE + j) * 8u + (lane >> 3);
        const uint32_t part = (lane & 7u) * 4u;
        if (rho < TROWS * KC) {
            const uint32_t i = rho >> 2, k = rho & 3u;
            const uint32_t rr = min(r0 + i, rows - 1);
            src[j] = A + (static_cast<size_t>(rr) * inner + k) * polyw + slot_base + part;
        } else {
            const uint32_t rb = rho - TROWS * KC, ch = rb & 1u, t = rb >> 1, k = t & 3u, u = t >> 2;
            const uint32_t c = ((u >> 2) << 3) + (ch << 2) + (u & 3u);
            const uint32_t cc = min(c0 + c, cols - 1);
            src[j] = B + (static_cast<size_t>(k) * cols + cc) * polyw + slot_base + part;
        }
    }
    auto issue = [&](uint32_t stage) {
#pragma unroll
        for (int j = 0; j < LOADS_PER_WAVE; ++j) {
            const uint32_t row0 = (wave * LOADS_PER_WAVE + j) * 8u;  // wave-uniform
            __builtin_amdgcn_global_load_lds(src[j], lds + stage * STAGE_WORDS + row0 * SLOTS, 16, 0, 0);
            src[j] += row0 < TROWS * KC ? stride_a : stride_b;
        }
    };

    uint64_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;

    const uint32_t nch = inner / KC;
    const uint32_t lazy = lc.lazy_terms;
    uint32_t pending = 0;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds));
    const uint32_t b_addr = lds0 + (A_WORDS + wq * 4u * KC * 2u * SLOTS + half * SLOTS + slot) * 4u;
    constexpr uint32_t STAGE_BYTES = STAGE_WORDS * 4u;

    // All of a chunk's LDS reads sit at its top (no registers are left to keep a second operand set in flight: 64
    // accumulators + 48 operand words; splitting the chunk into k-steps with two small sets made the compiler spill
    // and copy registers whose reads were still in flight).  The other three waves of the SIMD cover the wait.
    const uint32_t a_even = lds0 + (wr * 8u * KC * SLOTS + slot) * 4u, a_odd = a_even + SLOTS * 4u;
    issue(0);
    if (nch > 1) issue(1);
    if (nch > 2) issue(2);
    uint32_t stage = 0;
    for (uint32_t ch = 0; ch < nch; ++ch) {
        // chunk ch must have landed; up to two younger chunks stay in flight
        if (ch + 2 < nch) MMDMA_WAIT_VM(2 * LOADS_PER_WAVE);
        else if (ch + 1 < nch) MMDMA_WAIT_VM(LOADS_PER_WAVE);
        else MMDMA_WAIT_VM(0);
        asm volatile("s_barrier" ::: "memory");  // ... for every wave, and everyone is done with chunk ch-1
        if (ch + 3 < nch) issue((stage + 3) % STAGES);
        const uint32_t off = stage * STAGE_BYTES;
        uint64_t a[8][2], b[4][2];
        static_for<4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            b[j][0] = ds_read2st64<4 * j, 4 * j + 1>(b_addr + off);      // (k0, k1)
            b[j][1] = ds_read2st64<4 * j + 2, 4 * j + 3>(b_addr + off);  // (k2, k3)
        });
        static_for<8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            a[i][0] = ds_read2st64<2 * i, 2 * i + 1>(a_even + off);  // (k0, k2)
            a[i][1] = ds_read2st64<2 * i, 2 * i + 1>(a_odd + off);   // (k1, k3)
        });
#pragma unroll
        for (int j = 0; j < 4; ++j) lds_wait<0>(b[j][0], b[j][1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_wait<0>(a[i][0], a[i][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t bk[KC] = {static_cast<uint32_t>(b[j][0]), static_cast<uint32_t>(b[j][0] >> 32),
                                     static_cast<uint32_t>(b[j][1]), static_cast<uint32_t>(b[j][1] >> 32)};
#pragma unroll
            for (int k = 0; k < KC; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t av = (k & 2) ? static_cast<uint32_t>(a[i][k & 1] >> 32) : static_cast<uint32_t>(a[i][k & 1]);
                    acc[i][j] += static_cast<uint64_t>(av) * bk[k];
                }
        }
        pending += KC;
        if (pending + KC > lazy) {
            pending = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = reduce_u64_sum(acc[i][j], q, lc.mu64);
        }
        stage = (stage + 1) % STAGES;
    }
    const BoundedReduce br = bounded_reduce_setup(lc.kbits, lc.mu64, inner);
    // Round 4: the epilogue is a sixth of this kernel's instructions at an inner dimension of 64 (32 outputs per lane after
    // 64 MACs each); the output pointer is carried along the tile by two strides instead of being rebuilt from (r, c) with
    // 64-bit multiplies per output (11 -> ~4 VALU instructions per store).
    const uint32_t c_first = c0 + wq * 8u + half * 4u;
    uint32_t *const dst00 = C + (static_cast<size_t>(r0 + wr * 8u) * cols + c_first) * polyw + slot_base + slot;
    const size_t row_step = static_cast<size_t>(cols) * polyw;
    auto store_tile = [&](auto reduce) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (r0 + wr * 8 + i >= rows) continue;
            uint32_t *dst = dst00 + i * row_step;
#pragma unroll
            for (int j = 0; j < 4; ++j, dst += polyw) {
                if (c_first + j >= cols) continue;
                // non-temporal when C is far larger than the Infinity Cache (nt_c, set by the launcher from 1 GiB): it is
                // written once and must not displace the A / B panels their second reader still needs (-2..4 %); a small C
                // stays cacheable for whatever reads it next
                if (nt_c) __builtin_nontemporal_store(reduce(acc[i][j]), dst);
                else *dst = reduce(acc[i][j]);
            }
        }
    };
    if (br.ok) store_tile([&](uint64_t v) { return reduce_u64_bounded(v, q, br); });
    else store_tile([&](uint64_t v) { return reduce_u64_sum(v, q, lc.mu64); });
}
}  // namespace mmdma32

// 32 slots x 32 x 32 tile per workgroup; -1: shape not supported
int launch_matmul_dma32_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    GpuContext *ctx = out->ctx;
    const uint32_t rows = static_cast<uint32_t>(lhs->rows), inner = static_cast<uint32_t>(lhs->cols),
                   cols = static_cast<uint32_t>(rhs->cols);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out)), N = static_cast<uint32_t>(ctx->N);
    if (ctx->wide || N < mmdma32::SLOTS || (N % mmdma32::SLOTS) != 0 || inner < mmdma32::KC || (inner % mmdma32::KC) != 0) return -1;
    const uint32_t row_tiles = (rows + mmdma32::TROWS - 1) / mmdma32::TROWS, col_tiles = (cols + mmdma32::TCOLS - 1) / mmdma32::TCOLS;
    const uint32_t slot_chunks = N / mmdma32::SLOTS;
    const uint64_t groups = static_cast<uint64_t>(L) * slot_chunks;
    const uint64_t blocks = groups * row_tiles * col_tiles;
    if (blocks > 0x7fffffffull) return set_error("gpu_matrix_mul: matrix too large");
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(configured.load() & bit)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mmdma32::kernel_u32),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(mmdma32::LDS_BYTES)));
        configured.fetch_or(bit);
    }
    const uint32_t remap = (groups % 8 == 0) ? 1u : 0u;
    ctx->last_kernel = "mmdma32::kernel_u32 (32 slots x 32x32 tile, 16 waves, global_load_lds four-stage ring)";
    MXX_LAUNCH(mmdma32::kernel_u32, dim3(static_cast<unsigned>(blocks)), dim3(1024), mmdma32::LDS_BYTES, ctx->stream,
                       static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(lhs->data),
                       static_cast<const uint32_t *>(rhs->data), ctx->d_limbs, rows, inner, cols, L, N, row_tiles,
                       col_tiles, slot_chunks, remap, out->bytes >= (size_t(1) << 30) ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    return 0;
}

// -1: shape not supported (the caller falls back to the register-staged LDS kernel)
int launch_matmul_dma_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs) {
    GpuContext *ctx = out->ctx;
    const uint32_t rows = static_cast<uint32_t>(lhs->rows), inner = static_cast<uint32_t>(lhs->cols),
                   cols = static_cast<uint32_t>(rhs->cols);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out)), N = static_cast<uint32_t>(ctx->N);
    if (ctx->wide || N < 64 || (N % 64) != 0 || inner < mmdma::KC || (inner % mmdma::KC) != 0) return -1;
    const uint32_t row_tiles = (rows + mmdma::TROWS - 1) / mmdma::TROWS, col_tiles = (cols + mmdma::TCOLS - 1) / mmdma::TCOLS;
    const uint32_t slot_chunks = N / 64;
    const uint64_t groups = static_cast<uint64_t>(L) * slot_chunks;
    const uint64_t blocks = groups * row_tiles * col_tiles;
    if (blocks > 0x7fffffffull) return set_error("gpu_matrix_mul: matrix too large");
    static std::atomic<uint64_t> configured{0};
    const uint64_t bit = 1ull << (ctx->device & 63);
    if (!(configured.load() & bit)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mmdma::kernel_u32),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(mmdma::LDS_BYTES)));
        configured.fetch_or(bit);
    }
    const uint32_t remap = (groups % 8 == 0) ? 1u : 0u;
    ctx->last_kernel = "mmdma::kernel_u32 (64 slots x 32x16 tile, 8 waves, global_load_lds three-stage ring)";
    MXX_LAUNCH(mmdma::kernel_u32, dim3(static_cast<unsigned>(blocks)), dim3(512), mmdma::LDS_BYTES, ctx->stream,
                       static_cast<uint32_t *>(out->data), static_cast<const uint32_t *>(lhs->data),
                       static_cast<const uint32_t *>(rhs->data), ctx->d_limbs, rows, inner, cols, L, N, row_tiles,
                       col_tiles, slot_chunks, remap, out->bytes >= (size_t(1) << 30) ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    return 0;
}
