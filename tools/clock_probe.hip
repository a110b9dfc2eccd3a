// clock_probe.hip — the engine clock the chip SUSTAINS while every SIMD issues multiply-class VALU instructions.
// bench.py prices issue floors at the 2.4 GHz peak clock; tools/valu_rates.hip measured its per-opcode prices with one
// workgroup per CU.  This probe fills 1, 32 or all 256 CUs with 8 waves per SIMD of independent v_mul_hi_u32 / v_mad_u64_u32
// chains and reads both device timers around the loop: clock64() (s_memtime: engine-clock ticks) and wall_clock64()
// (s_memrealtime: constant 100 MHz).  ticks / real time = the clock the waves actually ran at; wave-instructions / ticks =
// cycles per instruction at that clock.  Round 5, one box: 2.22-2.37 GHz with all 256 CUs busy (2.39-2.41 with a few
// workgroups), v_mul_hi_u32 4.2 and v_add_u32 2.4 cycles-at-2.4-GHz per wave-instruction per SIMD from the kernel time -
// the table prices of profiles/r02_valu_rates.txt hold, and an issue floor priced at 2.4 GHz is 2-8 % optimistic.  The
// forward transform's butterfly as a stream of its own (five instructions + a mask on eight independent pairs): 24.4
// cycles per wave-butterfly, i.e. 4.07 per instruction, the full-rate opcodes included.
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define ITERS 20000
#define ILP 8

template <int OP>
__global__ void __launch_bounds__(512) probe(uint32_t seed, uint64_t *out, uint32_t *sink) {
    uint32_t v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = seed + threadIdx.x * 2654435761u + i * 40503u;
    const uint32_t m = seed | 1u;
    __syncthreads();
    const uint64_t t0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
            else if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
            else asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
        }
    }
    const uint64_t t1 = clock64(), r1 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc ^= v[i];
    if (acc == 0x12345678u) *sink = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = r1 - r0;
    }
}

// the forward transform's lazy butterfly (ntt_lds.h) on ILP independent pairs per thread with register twiddles:
// v_mul_hi_u32, v_mul_lo_u32, v_mad_u64_u32 (low word), v_sub_u32, v_add3_u32 - five instructions per butterfly
__global__ void __launch_bounds__(512, 8) bfly_probe(uint32_t seed, uint64_t *out, uint32_t *sink) {
    uint32_t u[ILP], v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        u[i] = (seed + threadIdx.x * 2654435761u + i * 40503u) & 0xffffffu;
        v[i] = (seed * 3u + threadIdx.x * 40503u + i * 7919u) & 0xffffffu;
    }
    uint32_t q = 16580609u, nq = 0u - q, w = 0u - 1234567u, ws = 319794563u, twoq = 2u * q;
    asm volatile("" : "+v"(nq), "+v"(w), "+v"(ws), "+v"(twoq));
    __syncthreads();
    const uint64_t t0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            const uint32_t V = v[i];
            const uint32_t nT = V * w + __umulhi(V, ws) * nq;
            const uint32_t U = u[i];
            u[i] = (U - nT) & 0x3ffffffu;  // the mask stands in for the occasional fold: keeps the values bounded
            v[i] = U + twoq + nT;
        }
    }
    const uint64_t t1 = clock64(), r1 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc ^= u[i] ^ v[i];
    if (acc == 0x12345678u) *sink = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t wv = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
        out[2 * wv] = t1 - t0;
        out[2 * wv + 1] = r1 - r0;
    }
}

static void run_bfly(int total_blocks) {
    const size_t waves = static_cast<size_t>(total_blocks) * 8;
    uint64_t *d = nullptr;
    uint32_t *sink = nullptr;
    hipMalloc(&d, waves * 16);
    hipMalloc(&sink, 4);
    hipLaunchKernelGGL(bfly_probe, dim3(total_blocks), dim3(512), 0, 0, 12345u, d, sink);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bfly_probe, dim3(total_blocks), dim3(512), 0, 0, 12345u, d, sink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bflies = static_cast<double>(ITERS / 4) * ILP * static_cast<double>(waves) / 1024.0;  // wave-butterflies per SIMD
    std::printf("lazy butterfly  %4d workgroups of 8 waves: kernel %.3f ms, %.2f cycles at 2.4 GHz per wave-butterfly per SIMD "
                "(five instructions + a mask; price sum 4.07 * 3 + 2.25 * 2 + 2.25 = 18.9 if v_add3 is full rate, 20.8 if not)\n",
                total_blocks, ms, ms * 1e-3 * 2.4e9 / bflies);
    hipFree(d);
    hipFree(sink);
}

template <int OP>
static void run(const char *name, int blocks, int waves_per_simd) {
    const int threads = 512;  // 8 waves: 2 per SIMD; blocks per CU = waves_per_simd / 2
    const int total_blocks = blocks * (waves_per_simd / 2);
    const size_t waves = static_cast<size_t>(total_blocks) * (threads / 64);
    uint64_t *d = nullptr;
    uint32_t *sink = nullptr;
    hipMalloc(&d, waves * 16);
    hipMalloc(&sink, 4);
    hipLaunchKernelGGL(probe<OP>, dim3(total_blocks), dim3(threads), 0, 0, 12345u, d, sink);  // warm-up
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<OP>, dim3(total_blocks), dim3(threads), 0, 0, 12345u, d, sink);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * waves);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (size_t w = 0; w < waves; ++w) {
        ticks += static_cast<double>(h[2 * w]);
        real += static_cast<double>(h[2 * w + 1]);
    }
    ticks /= waves;
    real /= waves;  // 100 MHz units
    const double ghz = ticks / (real * 10.0);  // ticks per nanosecond
    const double insts = static_cast<double>(ITERS) * ILP;
    // chip-level rate from the KERNEL's duration (the waves of a launch do not all run at once, so a wave's own loop time
    // says nothing about the SIMD's issue rate): wave-instructions per SIMD = insts * waves / (4 * CUs in use)
    const double simds = 4.0 * (blocks < 256 ? blocks : 256);
    const double per_simd = insts * static_cast<double>(waves) / simds;
    std::printf("%-14s %4d workgroups of 8 waves (%d per SIMD if spread evenly): kernel %.3f ms, a wave's loop %.1f us, engine clock %.3f GHz, "
                "%.2f cycles at 2.4 GHz per wave-instruction per SIMD (kernel time)\n",
                name, total_blocks, waves_per_simd, ms, real * 0.01, ghz, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int wps : {2, 4, 8, 32}) run<0>("v_mul_hi_u32", 256, wps);  // 32: four rounds of workgroups
    run<1>("v_mul_lo_u32", 256, 32);
    run<2>("v_add_u32", 256, 32);
    run_bfly(1024);
    run_bfly(4096);
    return 0;
}
