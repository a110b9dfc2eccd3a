// valu_rates.hip — issue cost of the VALU instructions a modular butterfly could be built from, on gfx950.
// Eight independent dependency chains per thread, 2 and 4 waves per SIMD, one workgroup per CU.  The number
// printed is cycles per wave-instruction per SIMD (wave64: 4.0 = 16 lanes per clock, 2.0 = 32 lanes per clock).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define ITERS 4096
#define ILP 8

#define OPS(X)                                                              \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1", 32)                            \
    X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1", 32)                            \
    X(2, "v_min_u32", "v_min_u32 %0, %0, %1", 32)                            \
    X(3, "v_and_b32", "v_and_b32 %0, %0, %1", 32)                            \
    X(4, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0", 32)                     \
    X(5, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1", 32)                    \
    X(6, "v_mul_hi_u32_u24", "v_mul_hi_u32_u24 %0, %0, %1", 32)              \
    X(7, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1", 32)                \
    X(8, "v_mad_i32_i24", "v_mad_i32_i24 %0, %0, %1, %1", 32)                \
    X(9, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 32)                      \
    X(10, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1", 32)                     \
    X(11, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 24", 32)             \
    X(12, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1", 32)              \
    X(13, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc", 32)              \
    X(14, "v_add_f32", "v_add_f32 %0, %0, %1", 32)                           \
    X(15, "v_mul_f32", "v_mul_f32 %0, %0, %1", 32)                           \
    X(16, "v_fmac_f32", "v_fmac_f32 %0, %1, %1", 32)                         \
    X(17, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1", 32)                       \
    X(18, "v_rndne_f32", "v_rndne_f32 %0, %0", 32)                           \
    X(19, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 32)                       \
    X(20, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0", 32)                       \
    X(21, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0", 32)                       \
    X(22, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1", 64)                     \
    X(23, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1", 64)                     \
    X(24, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %1", 64)                 \
    X(25, "v_add_f64", "v_add_f64 %0, %0, %1", 64)                           \
    X(26, "v_mul_f64", "v_mul_f64 %0, %0, %1", 64)                           \
    X(27, "v_fma_f64", "v_fma_f64 %0, %0, %1, %1", 64)                       \
    X(28, "v_mad_u64_u32", "v_mad_u64_u32 %0, vcc, %2, %2, %0", 64)          \
    X(29, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %1", 32)                     \
    X(30, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1", 32)               \
    X(31, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %1, %1", 32)                 \
    X(32, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %1, %1, %0", 32)               \
    X(33, "v_cmp+cndmask", "v_cmp_ge_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", 32) \
    X(34, "v_max_i32", "v_max_i32 %0, %0, %1", 32)                           \
    X(35, "v_sub_f32", "v_sub_f32 %0, %0, %1", 32)                           \
    X(36, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 17", 32)                        \
    X(37, "v_xor_b32", "v_xor_b32 %0, %0, %1", 32)                           \
    X(38, "v_add_co+addc", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc", 32)

template <int OP>
__global__ void k(uint32_t *out, uint32_t seed) {
    uint32_t a[ILP], b = seed + threadIdx.x;
    uint64_t w[ILP], b64 = (static_cast<uint64_t>(b) << 32) | seed;
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        a[i] = threadIdx.x * 7 + i + seed;
        w[i] = (static_cast<uint64_t>(a[i]) << 32) | a[i];
    }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
#define X(ID, NAME, ASM, WIDTH)                                                              \
    if (OP == ID) {                                                                          \
        if (WIDTH == 32) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(b) : "vcc");            \
        else asm volatile(ASM : "+v"(w[i]) : "v"(b64), "v"(b) : "vcc");                      \
    }
                OPS(X)
#undef X
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) r += a[i] + static_cast<uint32_t>(w[i]) + static_cast<uint32_t>(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static double clock_ghz = 2.4;

template <int OP>
double run(int wps, int insts) {
    const int blocks = 256, threads = 64 * 4 * wps;
    uint32_t *out;
    hipMalloc(&out, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(out, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(out, 12345);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(out);
    const double ops_per_simd = static_cast<double>(wps) * ITERS * 4 * ILP * insts;
    return ms * 1e-3 * clock_ghz * 1e9 / ops_per_simd;
}

int main() {
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    if (khz > 0) clock_ghz = khz * 1e-6;
    printf("clock used for cycle counts: %.2f GHz (device attribute; the sustained clock may be lower)\n", clock_ghz);
    printf("%-18s %10s %10s %10s\n", "instruction", "1 wave", "2 waves", "4 waves");
#define X(ID, NAME, ASM, WIDTH)                                                                        \
    {                                                                                                  \
        const int insts = (ID == 33 || ID == 38) ? 2 : 1;                                              \
        printf("%-18s %10.2f %10.2f %10.2f\n", NAME, run<ID>(1, insts), run<ID>(2, insts), run<ID>(4, insts)); \
    }
    OPS(X)
#undef X
    return 0;
}
