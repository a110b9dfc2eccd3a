// valu_rates.hip — measures per-CU issue rate of the integer / fp64 VALU ops the
// NTT and mat-mul kernels are built from (gfx950).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define UNROLL 16

template <int OP>
__global__ void rate_kernel(uint32_t *out, uint32_t seed) {
    uint32_t a[UNROLL], b = seed + threadIdx.x, c = seed * 3 + 1;
    uint64_t w[UNROLL];
    double d[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = threadIdx.x * 7 + i + seed; w[i] = a[i]; d[i] = a[i]; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));  // v_mul_lo_u32
            else if (OP == 1) a[i] = __umulhi(a[i], b);                     // v_mul_hi_u32
            else if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");  // v_mad_u64_u32
            else if (OP == 3) a[i] = __umul24(a[i], b);       // v_mul_u32_u24
            else if (OP == 4) a[i] = __umul24(a[i], b) + c;                  // v_mad_u32_u24
            else if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));  // v_add_u32
            else if (OP == 6) a[i] = min(a[i], a[i] - b);                    // v_sub + v_min
            else if (OP == 7) d[i] = fma(d[i], 1.0000001, 0.5);              // v_fma_f64
            else if (OP == 8) a[i] = __builtin_amdgcn_alignbit(a[i], b, 24);  // v_alignbit_b32
            else if (OP == 9) a[i] = (uint32_t)(((uint64_t)(a[i] & 0xffffffu) * (uint64_t)(b & 0xffffffu)) >> 32);     // v_mul_hi_u32_u24
            else if (OP == 10) a[i] = (a[i] >= b) ? a[i] - b : a[i];          // cmp + cndmask
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) r += a[i] + (uint32_t)w[i] + (uint32_t)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
double run(const char *name, int insts_per_op) {
    const int blocks = 256 * 8, threads = 256;
    uint32_t *out;
    hipMalloc(&out, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, threads>>>(out, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, threads>>>(out, 12345);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_insts = (double)blocks * (threads / 64) * ITERS * UNROLL;
    double per_cu_per_us = wave_insts / 256.0 / (ms * 1e3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: 4 SIMDs per CU
    double cyc = 2400.0 * 4.0 / per_cu_per_us;
    printf("%-28s %8.3f ms  %7.1f wave-ops/us/CU  ~%5.2f cyc/wave-op/SIMD (x%d insts)\n", name, ms, per_cu_per_us, cyc, insts_per_op);
    hipFree(out);
    return cyc;
}

int main() {
    run<5>("v_add_u32", 1);
    run<0>("v_mul_lo_u32", 1);
    run<1>("v_mul_hi_u32", 1);
    run<2>("v_mad_u64_u32", 1);
    run<3>("v_mul_u32_u24", 1);
    run<9>("v_mul_hi_u32_u24", 1);
    run<4>("v_mad_u32_u24", 1);
    run<6>("v_sub+v_min_u32", 2);
    run<10>("v_cmp+v_cndmask(+sub)", 3);
    run<8>("v_alignbit_b32", 1);
    run<7>("v_fma_f64", 1);
    return 0;
}
