// mac_rate_lds.hip — the 8x8 v_mad_u64_u32 register tile fed from LDS as in the mat-mul kernels (per 4 k: 8 A
// fragments + 8 B fragments of 16 bytes per lane), 2 waves per SIMD, no global traffic.  Compare with mac_rate.hip.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mac_rate_lds.hip -o tools/mac_rate_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(512) k(uint64_t *out, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2][16][64][4];  // [A|B][entry][slot][k] as in matmul_lds_kernel_u32
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = wave; e < 32; e += 8)
        for (int kk = 0; kk < 4; ++kk) lds[e >> 4][e & 15][lane][kk] = e * 977 + lane * 13 + kk;
    __syncthreads();
    const uint32_t wr = wave >> 2 & 1, wc = wave & 1;
    uint64_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 a4[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a4[i] = *reinterpret_cast<const uint4 *>(&lds[0][wr * 8 + i][lane][0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint4 b4 = *reinterpret_cast<const uint4 *>(&lds[1][wc * 8 + j][lane][0]);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].x) * b4.x;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].y) * b4.y;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].z) * b4.z;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] += static_cast<uint64_t>(a4[i].w) * b4.w;
        }
        if (MODE == 1) __syncthreads();  // one barrier per chunk, as in the kernels
        asm volatile("" ::: "memory");
    }
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) r += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name) {
    const int blocks = 256, threads = 512, iters = 1024;
    uint64_t *out;
    (void)hipMalloc(&out, blocks * threads * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms, %.2f cycles per MAC instruction per SIMD (2 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.03e9 / (2.0 * iters * 256));
    (void)hipFree(out);
}

int main() {
    run<0>("8x8 tile, operands re-read from LDS every 4 k");
    run<1>("... plus one workgroup barrier per 4 k");
    return 0;
}
