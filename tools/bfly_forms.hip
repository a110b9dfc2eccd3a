// bfly_forms.hip — cycles per wave for alternative formulations of the lazy NTT butterfly (gfx950).
// 256 blocks x (64*4*WPS) threads: every SIMD holds WPS waves; 8 independent butterflies per iteration.
// Build: hipcc --offload-arch=gfx950 -O3 tools/bfly_forms.hip -o tools/bfly_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 2048
#define ILP 8

template <int FORM>
__global__ void k(uint32_t *out, uint32_t seed, uint32_t q, uint32_t w, uint32_t ws) {
    uint32_t U[ILP], V[ILP];
    const uint32_t twoq = 2 * q, negq = 0u - q, negw = 0u - w;
#pragma unroll
    for (int i = 0; i < ILP; ++i) { U[i] = threadIdx.x * 7 + i + seed; V[i] = threadIdx.x * 13 + 3 * i + seed; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            uint32_t a, b, lo, hi, hq;
            uint64_t t;
            if (FORM == 0) {  // current: nT = V*(-w) + hi(V*w')*q ; A = U - nT ; B = U + 2q + nT
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(V[i]), "v"(negw));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi) : "v"(V[i]), "v"(ws));
                t = lo;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(hi), "s"(q) : "vcc");
                const uint32_t nT = static_cast<uint32_t>(t);
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(a) : "v"(U[i]), "v"(nT));
                asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(b) : "v"(U[i]), "s"(twoq), "v"(nT));
            } else if (FORM == 1) {  // no 3-operand ops at all
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(V[i]), "v"(w));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi) : "v"(V[i]), "v"(ws));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(hq) : "v"(hi), "s"(q));
                uint32_t T, u2;
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(T) : "v"(lo), "v"(hq));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "v"(U[i]), "v"(T));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(u2) : "s"(twoq), "v"(U[i]));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(b) : "v"(u2), "v"(T));
            } else if (FORM == 2) {  // signed: T = V*w - mulhi_i32(V, w')*q ; A = U + T ; B = U - T
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(V[i]), "v"(w));
                asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(hi) : "v"(V[i]), "v"(ws));
                t = lo;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(hi), "s"(negq) : "vcc");
                const uint32_t T = static_cast<uint32_t>(t);
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "v"(U[i]), "v"(T));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(b) : "v"(U[i]), "v"(T));
            } else if (FORM == 3) {  // signed, mul_lo + sub instead of the 64-bit mad
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(V[i]), "v"(w));
                asm volatile("v_mul_hi_i32 %0, %1, %2" : "=v"(hi) : "v"(V[i]), "v"(ws));
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(hq) : "v"(hi), "s"(q));
                uint32_t T;
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(T) : "v"(lo), "v"(hq));
                asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "v"(U[i]), "v"(T));
                asm volatile("v_sub_u32 %0, %1, %2" : "=v"(b) : "v"(U[i]), "v"(T));
            } else {  // FORM 4: the three multiplies alone
                asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(lo) : "v"(V[i]), "v"(w));
                asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(hi) : "v"(V[i]), "v"(ws));
                t = lo;
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t) : "v"(hi), "s"(q) : "vcc");
                a = static_cast<uint32_t>(t); b = U[i];
            }
            U[i] = a; V[i] = b;
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) r += U[i] ^ V[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int FORM>
void run(const char *name, int wps) {
    const int blocks = 256, threads = 64 * 4 * wps;
    uint32_t *out;
    (void)hipMalloc(&out, blocks * threads * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<FORM><<<blocks, threads>>>(out, 12345, 16580609u, 1234567u, 319797411u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<FORM><<<blocks, threads>>>(out, 12345, 16580609u, 1234567u, 319797411u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bfly_per_simd = (double)wps * ITERS * ILP;
    printf("%-34s waves/SIMD=%d  %7.3f ms  %5.2f cycles per butterfly per wave (2.03 GHz)\n", name, wps, ms, ms * 1e-3 * 2.03e9 / bfly_per_simd);
    (void)hipFree(out);
}

int main() {
    for (int wps : {2, 4}) {
        run<0>("F0 current (mad_u64, sub, add3)", wps);
        run<1>("F1 unsigned, VOP2 adds only", wps);
        run<2>("F2 signed (mad_u64, add, sub)", wps);
        run<3>("F3 signed (mul_lo, sub, add, sub)", wps);
        run<4>("F4 three multiplies only", wps);
    }
    return 0;
}
