// mac_rate.hip — v_mad_u64_u32 rate for the 8x8 register-tile pattern of the mat-mul kernels
// (64 64-bit accumulators, 8 a values x 8 b values per k), WPS waves per SIMD, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mac_rate.hip -o tools/mac_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int TR, int TC>
__global__ void __launch_bounds__(64 * 4 * (TR * TC > 32 ? 2 : 4)) k(uint64_t *out, uint32_t seed, int iters) {
    uint64_t acc[TR][TC];
    uint32_t a[TR], b[TC];
#pragma unroll
    for (int i = 0; i < TR; ++i) { a[i] = threadIdx.x * 7 + i + seed; 
#pragma unroll
        for (int j = 0; j < TC; ++j) acc[i][j] = i * j; }
#pragma unroll
    for (int j = 0; j < TC; ++j) b[j] = threadIdx.x * 3 + j + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int i = 0; i < TR; ++i) acc[i][j] += static_cast<uint64_t>(a[i]) * b[j];
#pragma unroll
            for (int i = 0; i < TR; ++i) a[i] += 0x9e3779b9u;  // keep the operands changing (VOP2 adds)
#pragma unroll
            for (int j = 0; j < TC; ++j) b[j] ^= a[j % TR];
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) r += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int TR, int TC>
void run(int wps) {
    const int blocks = 256, threads = 64 * 4 * wps, iters = 512;
    uint64_t *out;
    (void)hipMalloc(&out, blocks * threads * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<TR, TC><<<blocks, threads>>>(out, 1, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<TR, TC><<<blocks, threads>>>(out, 1, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double macs_per_simd = (double)wps * iters * 4 * TR * TC;
    printf("tile %dx%d waves/SIMD=%d: %.3f ms, %.2f cycles per MAC instruction per SIMD (incl. %d operand updates per %d MACs)\n", TR, TC, wps,
           ms, ms * 1e-3 * 2.03e9 / macs_per_simd, TR + TC, TR * TC);
    (void)hipFree(out);
}

int main() {
    for (int wps : {1, 2}) run<8, 8>(wps);
    for (int wps : {1, 2, 3, 4}) run<8, 4>(wps);
    for (int wps : {2, 4}) run<4, 4>(wps);
    return 0;
}
