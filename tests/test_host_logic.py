"""CPU-only: host-side logic of the mirror package (no GPU calls)."""
import hashlib

import pytest

import mxx_amd as mx
from mxx_amd import trapdoor as td
from oracle import oracle as O


@pytest.mark.parametrize("n,depth,bits", [(4, 2, 17), (128, 2, 16), (16384, 15, 24), (256, 3, 51), (1024, 5, 51)])
def test_product_basis_equals_oracle_basis(n, depth, bits):
    assert mx.gen_crt_basis(n, depth, bits) == O.gen_crt_basis(n, depth, bits)


def test_cpu_params_derived_quantities():
    p = mx.DCRTPolyParams(16384, 15, 24, 12)
    assert p.modulus_digits() == 30  # bench_matrix_mul: k = 15 * ceil(24/12)
    assert 15 * 23 < p.modulus_bits() <= 15 * 24
    assert p.decompose_last_mask() is None
    p2 = mx.DCRTPolyParams(16, 3, 17, 5)
    assert p2.modulus_digits() == 12 and p2.decompose_last_mask() == (1 << 2) - 1
    d = mx.DCRTPolyParams()
    assert (d.ring_dimension(), d.crt_depth(), d.crt_bits(), d.base_bits()) == (4, 2, 17, 1)
    assert d.modulus_digits() == 34  # gadget cols == size*modulus_bits for base 1 (gpu_dcrt_poly.rs:2038-2046)


def test_hash_seed_is_deterministic_and_domain_separated():
    k = bytes(range(32))
    s1 = mx.hash_seed_for_matrix(k, b"tag")
    s2 = mx.hash_seed_for_matrix(k, b"tag")
    s3 = mx.hash_seed_for_matrix(k, b"tag2")
    assert s1.to_bytes() == s2.to_bytes() != s3.to_bytes()
    msg = b"GpuDCRTPolyHashSampler/v2" + k + b"tag" + (0).to_bytes(4, "little")
    from mxx_amd.sampler import keccak256

    assert s1.to_bytes() == keccak256(msg)  # the default H is Keccak-256, as in the reference's tests (sampler/gpu.rs:267)
    assert mx.hash_seed_for_matrix(k, b"tag", "sha3_256").to_bytes() == hashlib.sha3_256(msg).digest()


def test_keccak256_known_answers():
    from mxx_amd.sampler import keccak256

    assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # two absorbed blocks (rate = 136 bytes) and the block-boundary paddings
    assert keccak256(b"a" * 135).hex() == "34367dc248bbd832f4e3e69dfaac2f92638bd0bbd18f2912ba4ef454919cf446"
    assert len({keccak256(b"a" * n) for n in (135, 136, 137, 272)}) == 4


def test_seed_word_layout():
    s = mx.GpuRngSeed.from_bytes(bytes(range(32)))
    assert s.words[0] == int.from_bytes(bytes(range(8)), "little")
    assert s.words[3] == int.from_bytes(bytes(range(24, 32)), "little")


def test_preimage_constants():
    # s = 1.8 (b+1) sigma^2 (sqrt(dnk)+sqrt(2n)+4.7), c = (b+1) sigma (trapdoor/gpu.rs:15-27)
    base, sigma, d, n, k = 1 << 12, 4.578, 1, 16384, 20
    c = td.preimage_c(base, sigma)
    s = td.preimage_smoothing_parameter(base, sigma, d, n, k)
    assert c == (base + 1.0) * sigma
    assert abs(s - 1.8 * (base + 1) * sigma * sigma * ((d * n * k) ** 0.5 + (2 * n) ** 0.5 + 4.7)) < 1e-6 * s
    assert s > c


def test_dist_type_ffi_codes():
    assert mx.DistType.FinRingDist().as_ffi() == 0
    assert mx.DistType.GaussDist(3.2).as_ffi() == 1
    assert mx.DistType.BitDist().as_ffi() == 2
    assert mx.DistType.TernaryDist().as_ffi() == 3


def test_block_offsets_and_bincode_block_framing():
    """host helpers of the stored-matrix reader (gpu_dcrt_poly.rs:1594-1641,1899-1909): no device involved"""
    from mxx_amd.matrix import _bincode_nested_bytes, _bincode_read_nested_bytes, block_offsets, block_size

    assert block_offsets(range(0, 5), 2) == [0, 2, 4, 5]
    assert block_offsets(range(3, 9), 3) == [3, 6, 9]
    assert block_offsets(range(0, 0), 4) == [0]
    assert block_size() == int(__import__("os").environ.get("BLOCK_SIZE", "100"))
    entries = [[b"", b"\x01" * 250, b"\x02" * 251], [b"\x03" * 70000]]  # lengths on both sides of bincode's 1 / 3 / 5-byte varints
    blob = _bincode_nested_bytes(entries)
    assert blob[:2] == bytes([2, 3]) and blob[2] == 0 and blob[3] == 250 and blob[254] == 251  # 251 -> tag byte, then u16
    assert _bincode_read_nested_bytes(blob) == entries


def test_rns_snapshot_validation_without_a_device():
    from mxx_amd.matrix import GpuDCRTMatrixRnsSnapshot, rns_bytes_len, rns_bytes_len_for_level

    class P:  # the two accessors the checks use
        def crt_depth(self):
            return 3

        def ring_dimension(self):
            return 16

    assert rns_bytes_len_for_level(P(), 0) == 128 and rns_bytes_len(P()) == 384
    with pytest.raises(AssertionError):
        rns_bytes_len_for_level(P(), 3)
    ok = GpuDCRTMatrixRnsSnapshot(2, 1, 1, False, 256, bytes(512))
    ok.validate_for_params(P())
    assert ok == GpuDCRTMatrixRnsSnapshot(2, 1, 1, False, 256, bytearray(512)) and ok != GpuDCRTMatrixRnsSnapshot(2, 1, 1, True, 256, bytes(512))
    for bad, msg in [
        (GpuDCRTMatrixRnsSnapshot(2, 1, 3, False, 512, bytes(1024)), "invalid RNS snapshot level"),
        (GpuDCRTMatrixRnsSnapshot(2, 1, 1, False, 264, bytes(528)), "bytes_per_poly mismatch"),
        (GpuDCRTMatrixRnsSnapshot(2, 1, 1, False, 256, bytes(511)), "byte length mismatch"),
    ]:
        with pytest.raises(AssertionError, match=msg):
            bad.validate_for_params(P())
