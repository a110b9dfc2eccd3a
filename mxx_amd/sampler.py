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


_KECCAK_RC = (
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
    0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
    0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
)
_KECCAK_ROT = ((0, 36, 3, 41, 18), (1, 44, 10, 45, 2), (62, 6, 43, 15, 61), (28, 55, 25, 21, 56), (27, 20, 39, 8, 14))
_M64 = (1 << 64) - 1


def keccak256(data: bytes) -> bytes:
    """Keccak-256 with the original 0x01 padding (what `keccak_asm::Keccak256` computes; hashlib's sha3_256 is the NIST
    variant with 0x06 padding).  The reference's tests instantiate the hash sampler with it (src/sampler/gpu.rs:267).
    Pure Python: it only ever hashes a key, a tag and a counter into a 32-byte seed."""
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]  # a[x][y]
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & _M64 if n else v
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i : off + 8 * i + 8], "little")
        for rc in _KECCAK_RC:
            c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
            d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
            a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
            b = [[0] * 5 for _ in range(5)]
            for x in range(5):
                for y in range(5):
                    b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], _KECCAK_ROT[x][y])
            a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
            a[0][0] ^= rc
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def _digest(hash_name: str, data: bytes) -> bytes:
    if hash_name in ("keccak256", "keccak_256"):
        return keccak256(data)
    return hashlib.new(hash_name, data).digest()


def hash_seed_for_matrix(key: bytes, tag: bytes, hash_name: str = "keccak256") -> GpuRngSeed:
    """H("GpuDCRTPolyHashSampler/v2" || key || tag || ctr_le32), src/sampler/gpu.rs:118-136; H is generic in the
    reference (its tests and callers use Keccak256), any hashlib name is accepted as well."""
    assert len(key) == 32
    out = b""
    counter = 0
    while len(out) < 32:
        out += _digest(hash_name, b"GpuDCRTPolyHashSampler/v2" + bytes(key) + bytes(tag) + (counter & 0xFFFFFFFF).to_bytes(4, "little"))
        counter += 1
    return GpuRngSeed.from_bytes(out[:32])


def sample_gpu_matrix_with_seed(params, nrow, ncol, dist: DistType, seed: GpuRngSeed) -> GpuDCRTPolyMatrix:
    if nrow == 0 or ncol == 0:
        return GpuDCRTPolyMatrix.zero(params, nrow, ncol)
    return GpuDCRTPolyMatrix.sample_distribution(params, nrow, ncol, dist.as_ffi(), dist.sigma, seed)


def sample_gpu_matrix_native(params, nrow, ncol, dist: DistType) -> GpuDCRTPolyMatrix:
    """a fresh seed per call (sampler/gpu.rs:144-151)"""
    return sample_gpu_matrix_with_seed(params, nrow, ncol, dist, random_gpu_rng_seed())


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
    """`PolyHashSampler<[u8;32]>` for the GPU (src/sampler/gpu.rs:48-116); H defaults to Keccak256, the hash the
    reference instantiates it with."""

    def __init__(self, hash_name: str = "keccak256"):
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


    def sample_hash_decomposed_columns(self, params, key, tag, nrow, total_ncol, col_start, col_len, dist):
        """trait default (src/sampler/mod.rs:84-97)"""
        return self.sample_hash_columns(params, key, tag, nrow, total_ncol, col_start, col_len, dist).decompose_owned()

    def sample_hash_small_decomposed_columns(self, params, key, tag, nrow, total_ncol, col_start, col_len, dist):
        """trait default (src/sampler/mod.rs:111-124)"""
        return self.sample_hash_columns(params, key, tag, nrow, total_ncol, col_start, col_len, dist).small_decompose_owned()
