"""mxx_amd — MI355X-native DCRT polynomial-matrix engine behind mxx's GPU C ABI.

The compute path is libgpupoly.so (hand-written HIP for gfx950, sources in
`mxx_amd/csrc`, ABI in `include/gpupoly.h`).  This package is the host-side mirror
of the reference's Rust wrappers for that path; it has no CPU fallback.
"""
from ._ffi import (  # noqa: F401
    GPU_MATRIX_DIST_BIT,
    GPU_MATRIX_DIST_GAUSS,
    GPU_MATRIX_DIST_TERNARY,
    GPU_MATRIX_DIST_UNIFORM,
    GPU_POLY_FORMAT_COEFF,
    GPU_POLY_FORMAT_EVAL,
    GpuPolyError,
    GpuRngSeed,
    detected_gpu_device_count,
    detected_gpu_device_ids,
    gpu_device_sync,
)
from .matrix import (  # noqa: F401
    GpuDCRTMatrixRnsSnapshot,
    GpuDCRTPolyMatrix,
    GpuP1CovarianceCache,
    block_offsets,
    block_size,
    one_rns_bytes,
    rns_bytes_len,
    rns_bytes_len_for_level,
)
from .params import DCRTPolyParams, GpuContext, GpuDCRTPolyParams, gen_crt_basis  # noqa: F401
from .poly import GpuDCRTPoly  # noqa: F401
from .sampler import (  # noqa: F401
    DistType,
    GpuDCRTPolyHashSampler,
    GpuDCRTPolyUniformSampler,
    hash_seed_for_matrix,
    keccak256,
    random_gpu_rng_seed,
    sample_gpu_matrix_native,
    sample_gpu_matrix_with_seed,
    sample_gpu_matrix_with_seed_columns,
)
from .trapdoor import (  # noqa: F401
    GpuDCRTPolyTrapdoorSampler,
    GpuDCRTTrapdoor,
    GpuPerturbationSamples,
    coeff_cached_matrix,
    compute_preimage_norm,
    p1_covariance_parameters,
    preimage_c,
    preimage_smoothing_parameter,
)
