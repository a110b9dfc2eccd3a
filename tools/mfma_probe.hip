// mfma_probe.hip — two questions behind the MFMA form of the fat R_q product (mxx_amd/csrc/matmul_mfma.hip):
//  (1) the lane map of v_mfma_i32_32x32x32_i8's A / B operands, checked with exact integer data;
//  (2) how fast a workgroup can stream "S consecutive slots of every polynomial" (runs of 4*S bytes at a
//      polynomial stride) when the workgroups that share a 128-byte line run on one XCD at about the same time.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void mfma_layout_kernel(const int8_t *a, const int8_t *b, int *d) {
    // assumed: lane l (r = l & 31, h = l >> 5) holds A[r][16h + j] and B[16h + j][r], j = 0..15
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v4i v; int8_t e[16]; } fa, fb;
    for (int j = 0; j < 16; ++j) {
        fa.e[j] = a[r * 32 + 16 * h + j];
        fb.e[j] = b[(16 * h + j) * 32 + r];
    }
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.v, fb.v, acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h, col = r;
        d[row * 32 + col] = acc[reg];
    }
}

__global__ void mfma16_layout_kernel(const int8_t *a, const int8_t *b, int *d) {
    // assumed: lane l (r = l & 31, h = l >> 5) holds A[r][8h + j] and B[8h + j][r], j = 0..7 (k = 16 per instruction)
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { long v; int8_t e[8]; } fa, fb;
    for (int j = 0; j < 8; ++j) {
        fa.e[j] = a[r * 16 + 8 * h + j];
        fb.e[j] = b[(8 * h + j) * 32 + r];
    }
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x16_i8(fa.v, fb.v, acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) d[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg];
}

__global__ void mfma_rate_kernel(int *out, int iters) {
    // issue rate of the two i8 forms, one wave per SIMD: 4 independent accumulators each
    v16i acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = threadIdx.x + i;
    v4i a32 = {1, 2, 3, 4}, b32 = {5, 6, 7, 8};
    long a16 = 0x0102030405060708l, b16 = 0x0807060504030201l;
    long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a32, b32, acc[i], 0, 0, 0);
    long t1 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x16_i8(a16, b16, acc[i], 0, 0, 0);
    long t2 = __builtin_amdgcn_s_memtime();
    int r = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) r += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[100000] = (int)(t1 - t0); out[100001] = (int)(t2 - t1); }
}

static void check_layout16() {
    std::vector<int8_t> a(32 * 16), b(16 * 32);
    for (int i = 0; i < 32 * 16; ++i) {
        a[i] = static_cast<int8_t>((i * 37 + 11) % 251 - 125);
        b[i] = static_cast<int8_t>((i * 91 + 5) % 241 - 120);
    }
    int8_t *da, *db;
    int *dd;
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dd, 4096 * 128);
    (void)hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
    mfma16_layout_kernel<<<1, 64>>>(da, db, dd);
    std::vector<int> d(1024);
    (void)hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int want = 0;
            for (int k = 0; k < 16; ++k) want += int(a[i * 16 + k]) * int(b[k * 32 + j]);
            if (want != d[i * 32 + j]) ++bad;
        }
    printf("mfma_i32_32x32x16_i8 lane map A[l&31][8(l>>5)+j], B[8(l>>5)+j][l&31]: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    const int iters = 2000;
    mfma_rate_kernel<<<1, 64>>>(dd, iters);
    std::vector<int> t(2);
    (void)hipMemcpy(t.data(), dd + 100000, 8, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per MFMA, one wave: 32x32x32_i8 %.1f, 32x32x16_i8 %.1f\n", t[0] / (4.0 * iters), t[1] / (4.0 * iters));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
}

static void check_layout() {
    std::vector<int8_t> a(32 * 32), b(32 * 32);
    for (int i = 0; i < 32 * 32; ++i) {
        a[i] = static_cast<int8_t>((i * 37 + 11) % 251 - 125);
        b[i] = static_cast<int8_t>((i * 91 + 5) % 241 - 120);
    }
    int8_t *da, *db;
    int *dd;
    (void)hipMalloc(&da, 1024); (void)hipMalloc(&db, 1024); (void)hipMalloc(&dd, 4096);
    (void)hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
    mfma_layout_kernel<<<1, 64>>>(da, db, dd);
    std::vector<int> d(1024);
    (void)hipMemcpy(d.data(), dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int want = 0;
            for (int k = 0; k < 32; ++k) want += int(a[i * 32 + k]) * int(b[k * 32 + j]);
            if (want != d[i * 32 + j]) ++bad;
        }
    printf("mfma_i32_32x32x32_i8 lane map A[l&31][16(l>>5)+j], B[16(l>>5)+j][l&31]: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
}

// ---- (2) run streaming --------------------------------------------------------------------------------
// data: [P polys][L limbs][N slots] u32.  Workgroup = (limb, chunk of S slots); it reads the S slots of every
// poly.  256 threads, 16 bytes per lane per load => 4096/(4S) polys per load instruction row.
template <int S>
__global__ void __launch_bounds__(256) run_stream_kernel(const uint32_t *__restrict__ data, uint32_t *__restrict__ out,
                                                         uint32_t P, uint32_t L, uint32_t N, uint32_t reread) {
    constexpr uint32_t PIECES = S / 4;        // 16-byte pieces per run
    constexpr uint32_t RUNS = 256 / PIECES;   // runs (polys) per load instruction
    const uint32_t chunks_per_limb = N / S, total = chunks_per_limb * L;
    // XCD-aware: blocks b, b+8, ... run on one XCD; give them consecutive chunks
    // (hardware places block id b on XCD b % 8); the `reread` readers of one chunk are neighbours on one XCD
    const uint32_t xcd = blockIdx.x & 7, idx = (blockIdx.x >> 3) / reread;
    const uint32_t per_xcd = total / 8;
    const uint32_t chunk = xcd * per_xcd + idx;
    if (chunk >= total) return;
    const uint32_t limb = chunk / chunks_per_limb, c = chunk % chunks_per_limb;
    const uint32_t piece = threadIdx.x % PIECES, run = threadIdx.x / PIECES;
    const uint4 *base = reinterpret_cast<const uint4 *>(data + (static_cast<size_t>(limb) * N + c * S + piece * 4));
    const size_t poly_stride16 = static_cast<size_t>(L) * N / 4;
    uint4 acc = {0, 0, 0, 0};
    for (uint32_t p = run; p < P; p += RUNS * 4) {
        uint4 v0 = base[p * poly_stride16];
        uint4 v1 = (p + RUNS < P) ? base[(p + RUNS) * poly_stride16] : uint4{0, 0, 0, 0};
        uint4 v2 = (p + 2 * RUNS < P) ? base[(p + 2 * RUNS) * poly_stride16] : uint4{0, 0, 0, 0};
        uint4 v3 = (p + 3 * RUNS < P) ? base[(p + 3 * RUNS) * poly_stride16] : uint4{0, 0, 0, 0};
        acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x; acc.y ^= v0.y ^ v1.y ^ v2.y ^ v3.y;
        acc.z ^= v0.z ^ v1.z ^ v2.z ^ v3.z; acc.w ^= v0.w ^ v1.w ^ v2.w ^ v3.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <int S>
static void run_stream(const uint32_t *data, uint32_t *out, uint32_t P, uint32_t L, uint32_t N, uint32_t reread) {
    const uint32_t blocks = (N / S) * L * reread;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    run_stream_kernel<S><<<blocks, 256>>>(data, out, P, L, N, reread);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0);
        run_stream_kernel<S><<<blocks, 256>>>(data, out, P, L, N, reread);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = double(P) * L * N * 4;
    printf("S=%2d slots (%3d-byte runs), %u workgroups, each line read by %u workgroup(s): %.3f ms  unique %.2f TB/s  delivered %.2f TB/s\n",
           S, S * 4, blocks, reread, best, bytes / best / 1e9, bytes * reread / best / 1e9);
}

int main(int argc, char **argv) {
    check_layout();
    check_layout16();
    if (argc > 1) return 0;  // layout / rate checks only
    const uint32_t P = 8192, L = 8, N = 16384;  // A and B of the 64^3 product at L = 8: 4.3 GB
    uint32_t *data, *out;
    const size_t words = size_t(P) * L * N;
    if (hipMalloc(&data, words * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, size_t(N / 4) * L * 4 * 256 * 4);
    (void)hipMemset(data, 1, words * 4);
    for (uint32_t reread = 1; reread <= 4; reread *= 2) {
        run_stream<4>(data, out, P, L, N, reread);
        run_stream<8>(data, out, P, L, N, reread);
        run_stream<16>(data, out, P, L, N, reread);
        run_stream<32>(data, out, P, L, N, reread);
        run_stream<64>(data, out, P, L, N, reread);
    }
    (void)hipFree(data); (void)hipFree(out);
    return 0;
}
