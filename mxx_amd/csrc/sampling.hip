// sampling.hip — seeded distribution sampling straight into device matrices.
// Replaces cuda/src/matrix/MatrixSampling.cu behind
// cuda/include/matrix/MatrixSampling.cuh:25-37.
//
// Every coefficient owns a ChaCha20 stream keyed by the GLOBAL polynomial index
// row*full_ncol + col (+1), the coefficient index (+1) and, for the uniform
// distribution, the limb index (+1): sample_distribution_columns therefore equals
// the matching column slice of the full sample for the same seed
// (reference: MatrixSampling.cu:232-289, test src/sampler/gpu.rs:323-361).
// One thread draws the integer once and writes its residue into every limb (the
// reference relaunches per limb and redraws); the result is transformed to EVAL.
#include "common.h"
#include "modarith.h"
#include "rng.h"

#include <algorithm>

static constexpr uint64_t kTagUniform = 0x6f70656e66686531ull;
static constexpr uint64_t kTagGauss = 0x6f70656e66686532ull;
static constexpr uint64_t kTagBit = 0x6f70656e66686533ull;
static constexpr uint64_t kTagTernary = 0x6f70656e66686534ull;

// uniform / bit / ternary: one short stream per coefficient (and per limb for uniform).  The
// HChaCha20 sub-keys depend on the limb only, so the block derives them once into LDS.
template <typename W>
__global__ void __launch_bounds__(256) sample_distribution_kernel(W *__restrict__ out, const LimbConst *__restrict__ limbs, size_t polys,
                                           size_t local_ncol, size_t full_ncol, size_t col_offset, uint32_t L,
                                           uint32_t N, int dist, GpuRngSeed seed) {
    __shared__ uint64_t ring[256 * RNG_RING_WORDS];
    __shared__ ChaChaKey keys[GPUPOLY_MAX_LIMBS];
    const bool uniform = dist == GPU_MATRIX_DIST_UNIFORM;
    if (threadIdx.x < (uniform ? L : 1u))
        keys[threadIdx.x] = uniform ? chacha_subkey(seed, static_cast<uint64_t>(threadIdx.x) + 1, kTagUniform)
                                    : chacha_subkey(seed, 0, dist == GPU_MATRIX_DIST_BIT ? kTagBit : kTagTernary);
    __syncthreads();
    const size_t idx = item_index();
    if (idx >= polys * N) return;
    const size_t p = idx / N;
    const uint32_t i = static_cast<uint32_t>(idx - p * N);
    const size_t row = p / local_ncol, lcol = p - row * local_ncol;
    const uint64_t gpoly = row * full_ncol + col_offset + lcol;
    W *dst = out + p * L * N + i;
    ChaChaRng rng;
    if (uniform) {
        for (uint32_t l = 0; l < L; ++l) {
            rng_init_keyed(rng, ring, keys[l], gpoly + 1, static_cast<uint64_t>(i) + 1);
            dst[static_cast<size_t>(l) * N] = static_cast<W>(rng_uniform_mod(rng, limbs[l].q));
        }
        return;
    }
    rng_init_keyed(rng, ring, keys[0], gpoly + 1, static_cast<uint64_t>(i) + 1);
    rng_fill(rng);
    int64_t z;
    if (dist == GPU_MATRIX_DIST_BIT) {
        z = static_cast<int64_t>(rng_next_u64(rng) & 1ull);
    } else {
        const uint64_t pick = rng_next_u64(rng) % 3ull;
        z = pick == 0 ? 0 : (pick == 1 ? 1 : -1);
    }
    for (uint32_t l = 0; l < L; ++l)
        dst[static_cast<size_t>(l) * N] = signed_to_residue_mu<W>(z, limbs[l].q, limbs[l].mu64);
}

// discrete Gaussian: persistent lanes (rng.h).  Wave w owns coefficients [w*64*per_lane, +64*per_lane);
// its lanes take them one at a time (wave_take), each coefficient with its own stream
// (sub-key shared by all).  Lanes finish at different times, so a lane's store is a lone 8 bytes:
// it goes to a compact int64 staging array ([poly][N]) that a coalesced pass expands into the L
// residues per coefficient (storing the residues from here cost a 32-byte HBM write per limb).
__global__ void __launch_bounds__(256) sample_gauss_kernel(int64_t *__restrict__ stage, size_t polys,
                                    uint32_t local_ncol, size_t full_ncol, size_t col_offset, uint32_t logN,
                                    double sigma, KarneyDivisor div, ChaChaKey key, uint32_t per_lane) {
    __shared__ uint64_t ring[256 * RNG_RING_WORDS];  // exactly 32 KB: five workgroups per CU
    const size_t total = polys << logN;  // polys < 2^32 (checked by the launcher)
    const size_t N = static_cast<size_t>(1) << logN;
    WaveChunk chunk = wave_chunk(total, per_lane);
    ChaChaRng rng;
    rng_init_keyed(rng, ring, key, 0, 0);
    KarneyFsm f;
    f.st = KS_DONE;
    bool have = false;
    size_t idx = 0;
    for (uint32_t step = 0;; ++step) {
        if ((step & 3) == 0) {
            if ((step & 7) == 0) {
                const bool take = f.st == KS_DONE;  // write the finished coefficient, open the next one's stream
                if (take && have) stage[idx] = f.result;
                const uint32_t e = wave_take(chunk, take);
                if (take) {
                    have = e < chunk.len;
                    if (have) {
                        idx = chunk.base + e;
                        const uint32_t p = static_cast<uint32_t>(idx >> logN);
                        const uint32_t row = p / local_ncol, lcol = p - row * local_ncol;
                        rng_reopen(rng, row * full_ncol + col_offset + lcol + 1, (idx & (N - 1)) + 1);
                        karney_begin(f, 0.0, sigma, div);
                    } else {
                        f.st = KS_IDLE;
                    }
                }
                if (__all(f.st == KS_IDLE)) break;
                if (f.st != KS_IDLE) rng_fill<10>(rng);
            }
            karney_heavy(f, rng);
        }
        karney_light(f, rng);
    }
}

// keep_coeff: leave the samples as coefficients (for a caller that decomposes them next) instead of finishing in EVAL
int sample_impl(GpuMatrix *out, int dist, double sigma, GpuRngSeed seed, size_t full_ncol, size_t col_offset, bool keep_coeff) {
    if (!out) return set_error("gpu_matrix_sample_distribution: null matrix");
    if (dist < GPU_MATRIX_DIST_UNIFORM || dist > GPU_MATRIX_DIST_TERNARY)
        return set_error("gpu_matrix_sample_distribution: invalid dist_type");
    if (dist == GPU_MATRIX_DIST_GAUSS && !(sigma > 0.0))
        return set_error("gpu_matrix_sample_distribution: sigma must be positive for Gaussian sampling");
    if (col_offset + out->cols > full_ncol)
        return set_error("gpu_matrix_sample_distribution_columns: column window out of range");
    GpuContext *ctx = out->ctx;
    out->format = keep_coeff ? GPU_POLY_FORMAT_COEFF : GPU_POLY_FORMAT_EVAL;
    const size_t polys = matrix_polys(out);
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    const size_t total = polys * static_cast<size_t>(ctx->N);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out));
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    if (dist == GPU_MATRIX_DIST_GAUSS) {
        // enough lanes to fill the chip first, then up to 16 coefficients per lane
        if (polys >> 32) return set_error("gpu_matrix_sample_distribution: too many polynomials");
        const uint32_t per_lane = sampler_per_lane(total, reinterpret_cast<const void *>(sample_gauss_kernel), ctx->device, ctx->env.sampler_per_lane);
        const unsigned blocks = static_cast<unsigned>((total + 256u * per_lane - 1) / (256u * per_lane));
        const KarneyDivisor div = karney_divisor(sigma);
        const ChaChaKey key = chacha_subkey(seed, 0, kTagGauss);
        void *stage = nullptr;
        if (ctx_alloc(ctx, total * sizeof(int64_t), &stage)) return 1;
        hipLaunchKernelGGL(sample_gauss_kernel, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<int64_t *>(stage), polys,
                           static_cast<uint32_t>(out->cols), full_ncol, col_offset, ctx->logN, sigma, div, key, per_lane);
        const hipError_t err = hipGetLastError();
        const int rc = err == hipSuccess ? launch_scatter_i64(out, static_cast<const int64_t *>(stage)) : 0;
        ctx_free(ctx, stage);
        HIP_TRY(err);
        if (rc) return rc;
    } else {
        const dim3 blocks = item_grid(total, 256);
        if (ctx->wide)
            hipLaunchKernelGGL(sample_distribution_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                               static_cast<uint64_t *>(out->data), ctx->d_limbs, polys, out->cols, full_ncol, col_offset,
                               L, N, dist, seed);
        else
            hipLaunchKernelGGL(sample_distribution_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                               static_cast<uint32_t *>(out->data), ctx->d_limbs, polys, out->cols, full_ncol, col_offset,
                               L, N, dist, seed);
    }
    HIP_TRY(hipGetLastError());
    if (keep_coeff) return 0;
    // samples are coefficients; callers always get EVAL (MatrixSampling.cu:463-469)
    return launch_ntt(ctx, out->data, polys * L, static_cast<int>(L), false);
}

extern "C" int gpu_matrix_sample_distribution(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpu_matrix_sample_distribution: null matrix");
    return sample_impl(out, dist_type, sigma, seed, out->cols, 0, false);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_sample_distribution_columns(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed,
                                                      size_t full_ncol, size_t col_offset) {
    ABI_GUARD_BEGIN
    return sample_impl(out, dist_type, sigma, seed, full_ncol, col_offset, false);
    ABI_GUARD_END
}
