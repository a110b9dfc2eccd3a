"""GPU parity for the compact wire format and staging bytes."""
import os

import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu


def _payload_of(blob):
    from mxx_amd.matrix import _bincode_read_varint as rd

    pos = 2
    fields = []
    for _ in range(6):
        v, pos = rd(blob, pos)
        fields.append(v)
    level, nrow, ncol, max_bits, bpc, plen = fields
    return blob[0], blob[1], level, nrow, ncol, max_bits, bpc, blob[pos : pos + plen]


@pytest.mark.parametrize("n,depth,bits,base", [(4, 2, 17, 1), (16, 3, 18, 6), (128, 2, 16, 4), (16, 2, 51, 17), (64, 5, 24, 12)])
@pytest.mark.parametrize("eval_format", [False, True])
def test_compact_bytes_match_oracle_and_roundtrip(gpu, oracle, n, depth, bits, base, eval_format):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    coeff = rand_matrix(oracle, 100, 2, 3, moduli, n)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, coeff, False)
    if eval_format:
        m.ntt_all_in_place()
    blob = m.to_compact_bytes()
    version, fmt, level, nrow, ncol, max_bits, bpc, payload = _payload_of(blob)
    assert (version, fmt, level, nrow, ncol) == (1, 1 if eval_format else 0, depth - 1, 2, 3)
    want_payload, want_bits, want_bpc = oracle.compact_payload(coeff, moduli)
    assert (max_bits, bpc) == (want_bits, want_bpc)
    assert payload == want_payload
    back = gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, blob)
    assert back.is_ntt == eval_format
    assert back == m
    assert m.is_ntt == eval_format  # to_compact_bytes works on a clone


def test_compact_bytes_small_signed_values_and_zero(gpu, oracle):
    """Gaussian-sized entries pack into a few bits; the zero matrix has an empty payload
    (gpu_dcrt_poly.rs:2050-2075 zero_compact_bytes round trip)."""
    n = 128
    p = make_params(gpu, oracle, n, 2, 17, 1)
    moduli = p.moduli()
    rng = np.random.default_rng(4)
    v = rng.integers(-9, 10, size=(3, 2, n))
    coeff = np.stack([np.mod(v, q).astype(np.uint64) for q in moduli], axis=-2)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, coeff, False)
    blob = m.to_compact_bytes()
    _, _, _, _, _, max_bits, bpc, payload = _payload_of(blob)
    assert max_bits == 5 and bpc == 1  # |x| <= 9 -> 4 magnitude bits + sign
    assert len(payload) == (3 * 2 * n * 5 + 7) // 8
    assert payload == oracle.compact_payload(coeff, moduli)[0]
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, blob) == m
    z = gpu.GpuDCRTPolyMatrix.zero(p, 2, 2)
    zb = z.to_compact_bytes()
    assert _payload_of(zb)[5:] == (0, 0, b"")
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, zb) == z


def test_compact_bytes_lower_level_and_large(gpu, oracle):
    n = 1024
    p = make_params(gpu, oracle, n, 3, 24, 12)
    moduli = p.moduli()
    coeff = rand_matrix(oracle, 101, 3, 4, moduli, n)[:, :, :2]  # level 1 of 2
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, coeff, False)
    blob = m.to_compact_bytes()
    back = gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, blob)
    assert back.level == 1 and back == m
    # property at size: decode(encode(x)) == x and the width never exceeds bits(Q)+1
    assert _payload_of(blob)[5] <= sum(q.bit_length() for q in moduli[:2]) + 1


def test_cpu_staging_bytes_and_trapdoor_bytes(gpu, oracle):
    p = make_params(gpu, oracle, 128, 2, 16, 4)
    moduli = p.moduli()
    x = rand_matrix(oracle, 102, 2, 2, moduli, 128)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, True)
    assert gpu.GpuDCRTPolyMatrix.from_cpu_staging_bytes(p, m.to_cpu_staging_bytes()) == m
    td = gpu.GpuDCRTTrapdoor.new(p, 2, 4.578)
    td2 = gpu.GpuDCRTTrapdoor.from_compact_bytes(p, td.to_compact_bytes())
    assert td2 == td and td2.a_mat_coeff == td.a_mat_coeff
    assert gpu.GpuDCRTTrapdoor.from_compact_bytes(p, td.to_compact_bytes()[:-1]) is None


def test_mul_decompose_extension_matches_chunked(gpu, oracle):
    import ctypes as C
    from mxx_amd import _ffi

    n, base = 64, 10
    p = make_params(gpu, oracle, n, 2, 30, base)
    moduli = p.moduli()
    k = p.modulus_digits()
    S = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 103, 2, 3 * k, moduli, n), True)
    B = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 104, 3, 5, moduli, n), True)
    os.environ["MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"] = "2"  # the reference's column-chunk loop
    try:
        want = S.mul_decompose(B)
    finally:
        del os.environ["MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"]
    assert S.mul_decompose(B) == want  # default: the one-call extension
    out = gpu.GpuDCRTPolyMatrix.new_empty(p, 2, 5)
    _ffi.check_status(_ffi.lib().gpupoly_matrix_mul_decompose(out.raw, S.raw, B.raw, base), "gpupoly_matrix_mul_decompose")
    assert out == want


def _signed_matrix(rng, shape, n, moduli, magnitude):
    v = rng.integers(-magnitude, magnitude + 1, size=shape + (n,), dtype=np.int64)
    return v, np.stack([np.mod(v, q).astype(np.uint64) for q in moduli], axis=-2)


@pytest.mark.parametrize("n,depth,bits,magnitude", [(256, 5, 24, 2 ** 22), (256, 5, 24, 2 ** 40), (256, 10, 24, 2 ** 45), (128, 12, 51, 2 ** 49),
                                                      (128, 12, 51, 2 ** 62), (64, 3, 51, 2 ** 62)])
def test_compact_store_fast_forms_equal_the_general_kernels_and_the_oracle(gpu, oracle, hip_env, n, depth, bits, magnitude):
    """round 5: the store runs fast-path-only kernels first (|x| < q_0 / 2, |x| < q_0 q_1 / 2, every further limb checked) and
    falls back to the kernels with the general Garner path when a coefficient raises the flag.  Both forms, the
    LDS-assembled pack included, against the CPU packing - magnitudes in the one-limb range, the two-limb range (24- and
    51-bit limbs, one and two words), and a matrix whose single large entry forces the rerun."""
    p = make_params(gpu, oracle, n, depth, bits, 12)
    moduli = p.moduli()
    rng = np.random.default_rng(depth * 1000 + bits)
    v, coeff = _signed_matrix(rng, (3, 4), n, moduli, magnitude)
    want_payload, want_bits, want_bpc = oracle.compact_payload(coeff, moduli)
    blobs = []
    for mode in ("fast", "general"):
        if mode == "general":
            hip_env.set("MXX_HIP_SERDE", "general")
        m = gpu.GpuDCRTPolyMatrix.from_rns(p, coeff, False)
        blob = m.to_compact_bytes()
        _, _, _, _, _, max_bits, bpc, payload = _payload_of(blob)
        assert (max_bits, bpc) == (want_bits, want_bpc) and payload == want_payload, mode
        assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, blob) == m
        blobs.append(blob)
    assert blobs[0] == blobs[1]
    hip_env.unset("MXX_HIP_SERDE")
    # one coefficient beyond both fast paths (a uniform residue vector): the flag sends the call to the general kernels
    big = coeff.copy()
    big[1, 2, :, 7] = rand_matrix(oracle, 7, 1, 1, moduli, n)[0, 0, :, 7]
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, big, False)
    blob = m.to_compact_bytes()
    wp, wb, wc = oracle.compact_payload(big, moduli)
    _, _, _, _, _, max_bits, bpc, payload = _payload_of(blob)
    assert (max_bits, bpc) == (wb, wc) and payload == wp
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, blob) == m


def test_compact_view_is_the_same_bytes_without_a_host_copy(gpu, oracle):
    """`into_compact_view`: the framed payload in this thread's pinned staging buffer (valid until the next call); a buffer that
    is too small for the payload is retried with the length the library reports"""
    from mxx_amd import matrix as M

    n = 1024
    p = make_params(gpu, oracle, n, 3, 24, 12)
    moduli = p.moduli()
    rng = np.random.default_rng(9)
    _, small = _signed_matrix(rng, (2, 3), n, moduli, 5)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, small, True)
    want = m.to_compact_bytes()
    view = m.clone().into_compact_view()
    assert isinstance(view, memoryview) and bytes(view) == want
    # uniform residues need bits(Q) + 1 per coefficient: more than the first attempt's buffer
    uni = rand_matrix(oracle, 33, 4, 6, moduli, n)
    u = gpu.GpuDCRTPolyMatrix.from_rns(p, uni, False)
    before = M._pinned_capacity()
    blob = bytes(u.clone().into_compact_view())
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, blob) == u
    assert _payload_of(blob)[7] == oracle.compact_payload(uni, moduli)[0]
    assert M._pinned_capacity() >= max(before, len(blob))
