// bfly_u64_f64.hip — 51-bit limbs: cycles per wave of the lazy Cooley-Tukey butterfly in 64-bit integer arithmetic (what
// ntt_lds.h compiles to for W = uint64_t: a 64 x 64 high product + two low products per Shoup multiplication) against a
// double-precision formulation (values held as exact integers in doubles; the product's low part from an FMA).
// Also checks the FP64 form against exact 128-bit arithmetic on random operands.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/bfly_u64_f64.hip -o tools/bfly_u64_f64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define ITERS 1024
#define ILP 8

// FORM 0: integer.  nT = V * (-w) + hi64(V * w') * q ; A = U - nT ; B = U + 2q + nT   (values < 2^58)
// FORM 1: FP64.     c = rint(V * winv) ; h = V * w ; l = fma(V, w, -h) ; T = fma(-c, q, h) + l  in (-q, q)
//                   A = U + T ; B = U - T  (|values| grow by q per stage; exact while below 2^53 - the transform folds
//                   once per pass)
template <int FORM>
__global__ void k(double *out, uint64_t seed, uint64_t q, uint64_t w, uint64_t ws, double qd, double wd, double winv) {
    if (FORM == 0) {
        uint64_t U[ILP], V[ILP];
        const uint64_t twoq = 2 * q, negw = 0ull - w;
#pragma unroll
        for (int i = 0; i < ILP; ++i) { U[i] = (threadIdx.x * 7 + i + seed) % q; V[i] = (threadIdx.x * 13 + 3 * i + seed) % q; }
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                const uint64_t nT = V[i] * negw + __umul64hi(V[i], ws) * q;
                const uint64_t a = U[i] - nT, b = U[i] + twoq + nT;
                U[i] = a & 0x3ffffffffffffffull;  // keep the chain bounded (stands in for the per-pass fold)
                V[i] = b & 0x3ffffffffffffffull;
            }
        }
        uint64_t r = 0;
#pragma unroll
        for (int i = 0; i < ILP; ++i) r += U[i] ^ V[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<double>(r);
    } else {
        double U[ILP], V[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) { U[i] = static_cast<double>((threadIdx.x * 7 + i + seed) % q); V[i] = static_cast<double>((threadIdx.x * 13 + 3 * i + seed) % q); }
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                const double c = rint(V[i] * winv);
                const double h = V[i] * wd;
                const double l = fma(V[i], wd, -h);
                const double T = fma(-c, qd, h) + l;
                const double a = U[i] + T, b = U[i] - T;
                U[i] = a;
                V[i] = (FORM == 2) ? b - qd * rint(b * (1.0 / 2251799813684737.0)) : b;  // FORM 2: a fold on every output (upper bound of the folding cost)
                if (FORM == 1 && (it & 7) == 7) {  // one fold per 8 stages
                    U[i] -= qd * rint(U[i] * (1.0 / 2251799813684737.0));
                    V[i] -= qd * rint(V[i] * (1.0 / 2251799813684737.0));
                }
            }
        }
        double r = 0;
#pragma unroll
        for (int i = 0; i < ILP; ++i) r += U[i] + V[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    }
}

__global__ void check(const uint64_t *a, const uint64_t *w, uint64_t q, int *bad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double qd = static_cast<double>(q), ad = static_cast<double>(a[i]), wd = static_cast<double>(w[i]);
    const double winv = wd / qd;
    const double c = rint(ad * winv), h = ad * wd, l = fma(ad, wd, -h);
    double T = fma(-c, qd, h) + l;  // in (-q, q) up to the quotient estimate's error
    if (T < 0) T += qd;
    if (T >= qd) T -= qd;
    if (T < 0) T += qd;
    const unsigned __int128 p = static_cast<unsigned __int128>(a[i]) * w[i];
    const uint64_t want = static_cast<uint64_t>(p % q);
    if (static_cast<uint64_t>(T) != want) atomicAdd(bad, 1);
}

template <int FORM>
static void run(const char *name, int wps) {
    const int blocks = 256, threads = 64 * 4 * wps;
    double *out;
    (void)hipMalloc(&out, sizeof(double) * blocks * threads);
    const uint64_t q = 2251799813684737ull, w = 1234567890123457ull;
    const uint64_t ws = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 64) / q);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<FORM><<<blocks, threads>>>(out, 12345, q, w, ws, (double)q, (double)w, (double)w / (double)q);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<FORM><<<blocks, threads>>>(out, 12345, q, w, ws, (double)q, (double)w, (double)w / (double)q);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bflies_per_simd = static_cast<double>(ITERS) * ILP * wps;
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f cycles per wave-butterfly per SIMD (2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / bflies_per_simd);
    (void)hipFree(out);
}

int main() {
    // exactness of the FP64 product on 1M random operand pairs below q
    const int n = 1 << 20;
    const uint64_t q = 2251799813684737ull;
    uint64_t *ha = (uint64_t *)malloc(8 * n), *hw = (uint64_t *)malloc(8 * n), *da, *dw;
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; ha[i] = s % q;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; hw[i] = s % q;
    }
    ha[0] = q - 1; hw[0] = q - 1; ha[1] = 0; hw[1] = q - 1; ha[2] = q - 1; hw[2] = 1;
    int *bad, hbad = 0;
    (void)hipMalloc(&da, 8 * n); (void)hipMalloc(&dw, 8 * n); (void)hipMalloc(&bad, 4);
    (void)hipMemcpy(da, ha, 8 * n, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw, 8 * n, hipMemcpyHostToDevice);
    (void)hipMemset(bad, 0, 4);
    check<<<n / 256, 256>>>(da, dw, q, bad, n);
    (void)hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
    printf("FP64 modular product vs 128-bit reference on %d random pairs mod %llu: %d mismatches\n", n, (unsigned long long)q, hbad);
    for (int wps : {1, 2, 4}) {
        run<0>("u64 integer lazy butterfly", wps);
        run<1>("f64 butterfly, fold per 8 stages", wps);
        run<2>("f64 butterfly, fold every output", wps);
    }
    return hbad != 0;
}
