// bfly_rates.hip — cost of one lazy NTT butterfly (per wave64, per SIMD) in several instruction forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2048
template <int FORM>
__global__ void __launch_bounds__(512) k(uint32_t *out, uint32_t seed, uint32_t q) {
    uint32_t v[16], w[8], ws[8];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 7 + i + seed;
    for (int i = 0; i < 8; ++i) { w[i] = seed * (i + 3) + threadIdx.x; ws[i] = seed * (i + 11) ^ threadIdx.x; }
    const uint32_t twoq = q + q;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            uint32_t U = v[b], V = v[b + 8];
            if (FORM == 0) {  // product form: mul_lo + mul_hi + mad64 + sub + add3
                uint32_t nT = V * w[b] + __umulhi(V, ws[b]) * q;
                v[b] = U - nT; v[b + 8] = U + twoq + nT;
            } else if (FORM == 1) {  // no multiplies at all (adds only, same count)
                uint32_t nT = (V + w[b]) + ((V ^ ws[b]) + q);
                v[b] = U - nT; v[b + 8] = U + twoq + nT;
            } else if (FORM == 2) {  // 24-bit multiplies
                uint32_t hi = (uint32_t)(((uint64_t)(V & 0xffffffu) * (uint64_t)(ws[b] & 0xffffffu)) >> 32);
                uint32_t lo = __umul24(V, ws[b]);
                uint32_t t = __builtin_amdgcn_alignbit(hi, lo, 24);
                uint32_t T = __umul24(V, w[b]) - __umul24(t, q);
                v[b] = U + T; v[b + 8] = U + twoq - T;
            } else if (FORM == 3) {  // fp64: p = V*w exact, qhat = rint(p*qinv), r = fma(-qhat,q,p)
                double p = (double)V * (double)w[b];
                double qh = __builtin_rint(p * 1e-7);
                int32_t T = (int32_t)__builtin_fma(-qh, (double)q, p);
                v[b] = U + T; v[b + 8] = U + twoq - T;
            }
        }
        // rotate roles so values stay live and dependent across iterations
        uint32_t t0 = v[0];
#pragma unroll
        for (int i = 0; i < 15; ++i) v[i] = v[i + 1];
        v[15] = t0;
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int FORM> void run(const char *name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, threads = 512;
    uint32_t *out; hipMalloc(&out, blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<FORM><<<blocks, threads>>>(out, 12345, 16580609u); hipDeviceSynchronize();
    hipEventRecord(e0); k<FORM><<<blocks, threads>>>(out, 12345, 16580609u); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bf = (double)blocks * (threads / 64) * ITERS * 8;  // wave-butterflies
    double ns_per = ms * 1e6 / (bf / 1024.0);                 // per SIMD
    printf("%-34s %2d WG/CU  %7.3f ms  %6.2f ns per wave-butterfly per SIMD  (= %5.1f cyc @2.0GHz)\n", name, blocks_per_cu, ms, ns_per, ns_per * 2.0);
    hipFree(out);
}
int main() {
    for (int occ : {1, 2, 4}) {
        run<0>("mul_lo+mul_hi+mad64+sub+add3", occ);
        run<1>("adds only (5 ops)", occ);
        run<2>("u24 path (9 ops)", occ);
        run<3>("fp64 path", occ);
    }
    return 0;
}
