"""Distribution / hash samplers on the GPU (src/sampler/gpu.rs:16-253, src/sampler/mod.rs:12-120)."""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass

from . import _ffi
from ._ffi import GpuRngSeed
from .matrix import GpuDCRTPolyMatrix
from .poly import GpuDCRTPoly


@dataclass(frozen=True)
class DistType:
    """`DistType` (src/sampler/mod.rs:12-26)."""

    kind: str
    sigma: float = 0.0

    @staticmethod
    def FinRingDist():
        return DistType("fin_ring")

    @staticmethod
    def GaussDist(sigma: float):
        return DistType("gauss", float(sigma))

    @staticmethod
    def BitDist():
        return DistType("bit")

    @staticmethod
    def TernaryDist():
        return DistType("ternary")

    def as_ffi(self) -> int:
        return {
            "fin_ring": _ffi.GPU_MATRIX_DIST_UNIFORM,
            "gauss": _ffi.GPU_MATRIX_DIST_GAUSS,
            "bit": _ffi.GPU_MATRIX_DIST_BIT,
            "ternary": _ffi.GPU_MATRIX_DIST_TERNARY,
        }[self.kind]


_seed_source = None  # test-only hook: a callable returning 32 bytes per draw (see seed_source)


def random_gpu_rng_seed() -> GpuRngSeed:
    """OS randomness, as the reference (src/sampler/gpu.rs:138-142)."""
    if _seed_source is not None:
        return GpuRngSeed.from_bytes(_seed_source())
    return GpuRngSeed.from_bytes(os.urandom(32))


class seed_source:
    """Test-only: `with seed_source(iterable_of_32_byte_seeds):` makes every seed the samplers would
    draw from the OS come from the iterable instead, in call order, so a whole trapdoor / preimage
    chain can be replayed against the CPU restatement.  Not thread-safe; production code never sets it."""

    def __init__(self, seeds):
        self._it = iter(seeds)

    def __enter__(self):
        global _seed_source
        self._prev = _seed_source
        _seed_source = lambda: next(self._it)
        return self

    def __exit__(self, *exc):
        global _seed_source
        _seed_source = self._prev
        return False


def hash_seed_for_matrix(key: bytes, tag: bytes, hash_name: str = "sha3_256") -> GpuRngSeed:
    """H("GpuDCRTPolyHashSampler/v2" || key || tag || ctr_le32), src/sampler/gpu.rs:118-136."""
    assert len(key) == 32
    out = b""
    counter = 0
    while len(out) < 32:
        h = hashlib.new(hash_name)
        h.update(b"GpuDCRTPolyHashSampler/v2")
        h.update(key)
        h.update(tag)
        h.update((counter & 0xFFFFFFFF).to_bytes(4, "little"))
        out += h.digest()
        counter += 1
    return GpuRngSeed.from_bytes(out[:32])


def sample_gpu_matrix_with_seed(params, nrow, ncol, dist: DistType, seed: GpuRngSeed) -> GpuDCRTPolyMatrix:
    if nrow == 0 or ncol == 0:
        return GpuDCRTPolyMatrix.zero(params, nrow, ncol)
    return GpuDCRTPolyMatrix.sample_distribution(params, nrow, ncol, dist.as_ffi(), dist.sigma, seed)


def sample_gpu_matrix_with_seed_columns(params, nrow, total_ncol, col_start, col_len, dist, seed):
    if nrow == 0 or col_len == 0:
        return GpuDCRTPolyMatrix.zero(params, nrow, col_len)
    return GpuDCRTPolyMatrix.sample_distribution_columns(
        params, nrow, total_ncol, col_start, col_len, dist.as_ffi(), dist.sigma, seed
    )


class GpuDCRTPolyUniformSampler:
    """`PolyUniformSampler` for the GPU (src/sampler/gpu.rs:16-46)."""

    def sample_uniform(self, params, nrow, ncol, dist: DistType) -> GpuDCRTPolyMatrix:
        return sample_gpu_matrix_with_seed(params, nrow, ncol, dist, random_gpu_rng_seed())

    def sample_poly(self, params, dist: DistType) -> GpuDCRTPoly:
        return self.sample_uniform(params, 1, 1, dist).entry(0, 0)


class GpuDCRTPolyHashSampler:
    """`PolyHashSampler<[u8;32]>` for the GPU (src/sampler/gpu.rs:48-116); H defaults to Keccak-family sha3_256."""

    def __init__(self, hash_name: str = "sha3_256"):
        self.hash_name = hash_name

    def sample_hash(self, params, key: bytes, tag: bytes, nrow, ncol, dist: DistType) -> GpuDCRTPolyMatrix:
        return sample_gpu_matrix_with_seed(params, nrow, ncol, dist, hash_seed_for_matrix(key, tag, self.hash_name))

    def sample_hash_columns(self, params, key, tag, nrow, total_ncol, col_start, col_len, dist):
        seed = hash_seed_for_matrix(key, tag, self.hash_name)
        return sample_gpu_matrix_with_seed_columns(params, nrow, total_ncol, col_start, col_len, dist, seed)

    def sample_hash_decomposed(self, params, key, tag, nrow, ncol, dist):
        """== sample_hash(...).decompose() (src/sampler/gpu.rs:91-103), in one extension call."""
        seed = hash_seed_for_matrix(key, tag, self.hash_name)
        return GpuDCRTPolyMatrix.sample_distribution_decomposed(params, nrow, ncol, dist.as_ffi(), dist.sigma, seed)

    def sample_hash_small_decomposed(self, params, key, tag, nrow, ncol, dist):
        """== sample_hash(...).small_decompose() (src/sampler/gpu.rs:104-115), in one extension call."""
        seed = hash_seed_for_matrix(key, tag, self.hash_name)
        return GpuDCRTPolyMatrix.sample_distribution_decomposed(params, nrow, ncol, dist.as_ffi(), dist.sigma, seed, True)
