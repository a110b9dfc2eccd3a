// valu_latency.hip — how many resident waves per SIMD does gfx950 need to reach the VALU issue
// rate, for dependent chains (ILP=1) and independent streams (ILP=8)?  One workgroup of
// 64*WPS*4 threads per CU, so every SIMD holds exactly WPS waves.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_latency.hip -o tools/valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 8192

template <int OP, int ILP>
__global__ void k(uint32_t *out, uint32_t seed) {
    uint32_t a[ILP], b = seed + threadIdx.x;
    uint64_t w[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) { a[i] = threadIdx.x * 7 + i + seed; w[i] = a[i]; }
    for (int it = 0; it < ITERS / ILP; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(b) : "vcc");
                else if (OP == 4) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
                else if (OP == 5) {  // the 5-instruction lazy CT butterfly on (a[i], w[i].lo)
                    uint32_t lo, hi; uint64_t t;
                    asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(b));
                    asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(seed));
                    t = lo;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(hi), "v"(b) : "vcc");
                    uint32_t nT = (uint32_t)t, u = (uint32_t)w[i];
                    asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a[i]) : "v"(u), "v"(nT));
                    asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(u) : "v"(u), "v"(nT), "v"(b));
                    w[i] = u;
                }
            }
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) r += a[i] + (uint32_t)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP, int ILP>
void run(const char *name, int wps, int insts) {
    const int blocks = 256, threads = 64 * 4 * wps;
    uint32_t *out;
    hipMalloc(&out, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, ILP><<<blocks, threads>>>(out, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP, ILP><<<blocks, threads>>>(out, 12345);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x ITERS*8 ops; cycles at 2.03 GHz effective
    double ops_per_simd = (double)wps * ITERS * 8 * insts;
    double cyc = ms * 1e-3 * 2.03e9 / ops_per_simd;
    printf("%-14s ILP=%d waves/SIMD=%d  %7.3f ms  %5.2f cyc per wave-inst per SIMD\n", name, ILP, wps, ms, cyc);
    hipFree(out);
}

template <int OP> void sweep(const char *name, int insts) {
    for (int wps : {1, 2, 3, 4}) run<OP, 1>(name, wps, insts);
    for (int wps : {1, 2, 3, 4}) run<OP, 8>(name, wps, insts);
}

int main() {
    sweep<0>("v_add_u32", 1);
    sweep<4>("v_add3_u32", 1);
    sweep<1>("v_mul_lo_u32", 1);
    sweep<2>("v_mul_hi_u32", 1);
    sweep<3>("v_mad_u64_u32", 1);
    sweep<5>("butterfly(5)", 5);
    return 0;
}
