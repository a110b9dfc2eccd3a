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

static constexpr uint64_t kTagUniform = 0x6f70656e66686531ull;
static constexpr uint64_t kTagGauss = 0x6f70656e66686532ull;
static constexpr uint64_t kTagBit = 0x6f70656e66686533ull;
static constexpr uint64_t kTagTernary = 0x6f70656e66686534ull;

template <typename W>
__global__ void __launch_bounds__(256) sample_distribution_kernel(W *__restrict__ out, const LimbConst *__restrict__ limbs, size_t polys,
                                           size_t local_ncol, size_t full_ncol, size_t col_offset, uint32_t L,
                                           uint32_t N, int dist, double sigma, GpuRngSeed seed) {
    __shared__ uint64_t ring[256 * RNG_RING_WORDS];
    const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (idx >= polys * N) return;
    const size_t p = idx / N;
    const uint32_t i = static_cast<uint32_t>(idx - p * N);
    const size_t row = p / local_ncol, lcol = p - row * local_ncol;
    const uint64_t gpoly = row * full_ncol + col_offset + lcol;
    W *dst = out + p * L * N + i;
    ChaChaRng rng;
    if (dist == GPU_MATRIX_DIST_UNIFORM) {
        for (uint32_t l = 0; l < L; ++l) {
            rng_init(rng, ring, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, static_cast<uint64_t>(l) + 1, kTagUniform);
            dst[static_cast<size_t>(l) * N] = static_cast<W>(rng_uniform_mod(rng, limbs[l].q));
        }
        return;
    }
    int64_t z;
    if (dist == GPU_MATRIX_DIST_GAUSS) {
        rng_init(rng, ring, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, 0, kTagGauss);
        z = sample_integer_karney(rng, 0.0, sigma);
    } else if (dist == GPU_MATRIX_DIST_BIT) {
        rng_init(rng, ring, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, 0, kTagBit);
        rng_fill(rng);
        z = static_cast<int64_t>(rng_next_u64(rng) & 1ull);
    } else {
        rng_init(rng, ring, seed, gpoly + 1, static_cast<uint64_t>(i) + 1, 0, kTagTernary);
        rng_fill(rng);
        const uint64_t pick = rng_next_u64(rng) % 3ull;
        z = pick == 0 ? 0 : (pick == 1 ? 1 : -1);
    }
    for (uint32_t l = 0; l < L; ++l)
        dst[static_cast<size_t>(l) * N] = signed_to_residue<W>(z, static_cast<W>(limbs[l].q));
}

static int sample_impl(GpuMatrix *out, int dist, double sigma, GpuRngSeed seed, size_t full_ncol, size_t col_offset) {
    if (!out) return set_error("gpu_matrix_sample_distribution: null matrix");
    if (dist < GPU_MATRIX_DIST_UNIFORM || dist > GPU_MATRIX_DIST_TERNARY)
        return set_error("gpu_matrix_sample_distribution: invalid dist_type");
    if (dist == GPU_MATRIX_DIST_GAUSS && !(sigma > 0.0))
        return set_error("gpu_matrix_sample_distribution: sigma must be positive for Gaussian sampling");
    if (col_offset + out->cols > full_ncol)
        return set_error("gpu_matrix_sample_distribution_columns: column window out of range");
    GpuContext *ctx = out->ctx;
    out->format = GPU_POLY_FORMAT_EVAL;
    const size_t polys = matrix_polys(out);
    if (polys == 0) return 0;
    if (ctx_activate(ctx)) return 1;
    const size_t total = polys * static_cast<size_t>(ctx->N);
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out));
    const unsigned blocks = static_cast<unsigned>((total + 255) / 256);
    if (ctx->wide)
        hipLaunchKernelGGL(sample_distribution_kernel<uint64_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<uint64_t *>(out->data), ctx->d_limbs, polys, out->cols, full_ncol, col_offset, L,
                           (uint32_t)ctx->N, dist, sigma, seed);
    else
        hipLaunchKernelGGL(sample_distribution_kernel<uint32_t>, dim3(blocks), dim3(256), 0, ctx->stream,
                           static_cast<uint32_t *>(out->data), ctx->d_limbs, polys, out->cols, full_ncol, col_offset, L,
                           (uint32_t)ctx->N, dist, sigma, seed);
    HIP_TRY(hipGetLastError());
    // samples are coefficients; callers always get EVAL (MatrixSampling.cu:463-469)
    return launch_ntt(ctx, out->data, polys * L, static_cast<int>(L), false);
}

extern "C" int gpu_matrix_sample_distribution(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpu_matrix_sample_distribution: null matrix");
    return sample_impl(out, dist_type, sigma, seed, out->cols, 0);
    ABI_GUARD_END
}

extern "C" int gpu_matrix_sample_distribution_columns(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed,
                                                      size_t full_ncol, size_t col_offset) {
    ABI_GUARD_BEGIN
    return sample_impl(out, dist_type, sigma, seed, full_ncol, col_offset);
    ABI_GUARD_END
}
