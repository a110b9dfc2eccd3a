// ntt_phases.hip — diagnostic: where does a wave of the forward LDS NTT spend its cycles?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNTT_STAMPS -Imxx_amd/csrc tools/ntt_phases.hip mxx_amd/libgpupoly.so
#include "ntt_lds.h"
#include <cstdio>
#include <vector>

static bool is_prime(uint64_t n) { for (uint64_t d = 2; d * d <= n; ++d) if (n % d == 0) return false; return n > 1; }

int main() {
    const uint32_t logN = 14, N = 1u << logN, L = 4;
    std::vector<uint64_t> moduli;
    for (uint64_t q = (1ull << 24) + 1 - 2 * N; moduli.size() < L; q -= 2 * N) if (is_prime(q)) moduli.push_back(q);
    int gid = 0;
    GpuContext *ctx = nullptr;
    if (gpu_context_create(logN, L - 1, 1, moduli.data(), L, &gid, 1, &ctx)) { printf("ctx: %s\n", gpu_last_error()); return 1; }
    const size_t vectors = 4096;
    uint32_t *data;
    hipMalloc(&data, vectors * N * 4);
    hipMemset(data, 1, vectors * N * 4);
    auto kern = ntt_fwd_lazy_kernel<uint32_t, 14, 5, 3>;
    const size_t lds = lds_padded_words(N) * 4;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) {
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) { hipMalloc(&dbuf, vectors * 8 * 8 * 8); hipMemcpyToSymbol(HIP_SYMBOL(g_ntt_stamps), &dbuf, sizeof(dbuf)); }
        hipMemset(dbuf, 0, vectors * 8 * 8 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(512), dim3(512), lds, 0, data, (const TwPair<uint32_t> *)ctx->d_tw2_fwd, ctx->d_limbs, L, (uint32_t)vectors);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> all(vectors * 8 * 8);
        hipError_t ee = hipMemcpy(all.data(), dbuf, all.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long st[16] = {0};
        for (size_t w = 0; w < vectors * 8; ++w) for (int i = 0; i < 8; ++i) st[i] += all[w * 8 + i];
        if (ee != hipSuccess) printf("memcpyFromSymbol: %s\n", hipGetErrorString(ee));
        ee = hipGetLastError(); if (ee != hipSuccess) printf("last: %s\n", hipGetErrorString(ee));
        const double waves = 512 * 8.0 ;
        const char *names[8] = {"hbm load wait", "pass0 bfly", "lds wr+barrier", "pass1", "lds wr+barrier", "pass2", "barrier", "copy-out"};
        double tot = 0; for (int i = 0; i < 8; ++i) tot += st[i] / waves;
        printf("kernel %.1f us (stamped build); per-wave cycles (s_memtime ticks):\n", ms * 1e3);
        for (int i = 0; i < 8; ++i) printf("  %-16s %9.0f  %5.1f%%\n", names[i], st[i] / waves, 100.0 * st[i] / waves / tot);
        printf("  total %.0f\n", tot);
    }
    return 0;
}
