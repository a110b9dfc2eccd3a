"""The reference's own GPU unit tests, replayed one for one through the host mirror (same test names, same parameter
sets, same values, same predicates):

  src/matrix/gpu_dcrt_poly.rs:1962-2721    test_gpu_matrix_*
  src/poly/dcrt/gpu.rs:1219-1420           test_gpu_dcrtpoly_*
  src/sampler/gpu.rs:280-400               test_gpu_*sampler*, test_sample_gpu_matrix_with_seed_gauss_coeff_lt_6sigma
  src/sampler/trapdoor/gpu.rs:547-880      test_gpu_trapdoor_*, test_gpu_preimage_*, test_gpu_p_hat_*

Where the reference draws operands from the CPU samplers (OpenFHE) the replay draws them from the GPU samplers - the
predicates do not depend on where the operands come from.  The two cross-device tests run with two contexts on one
device when the box has a single GPU (the reference returns early there)."""
import math
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIGMA = 4.578


def cpu_params(n, depth, bits, base):
    from mxx_amd.params import DCRTPolyParams

    return DCRTPolyParams(n, depth, bits, base)


def gpu_params_from_cpu(gpu, p):
    moduli, _bits, _depth = p.to_crt()
    return gpu.GpuDCRTPolyParams(p.ring_dimension(), moduli, p.base_bits())


@pytest.fixture()
def P(gpu):  # gpu_test_params() of gpu_dcrt_poly.rs:1946 and poly/dcrt/gpu.rs:1206
    return gpu_params_from_cpu(gpu, cpu_params(128, 2, 17, 1))


@pytest.fixture()
def PS(gpu):  # gpu_test_params() of sampler/gpu.rs:270 and sampler/trapdoor/gpu.rs:496
    return gpu_params_from_cpu(gpu, cpu_params(128, 2, 16, 8))


def gpu_test_seed(gpu, base, offset):
    b = bytearray(32)
    b[:8] = ((base + offset) & 0xFFFFFFFFFFFFFFFF).to_bytes(8, "little")
    return gpu.GpuRngSeed.from_bytes(bytes(b))


def const(gpu, p, v):
    return gpu.GpuDCRTPoly.from_usize_to_constant(p, v)


def zero(gpu, p):
    return gpu.GpuDCRTPoly.const_zero(p)


# ---------------------------------------------------------------------------------- src/matrix/gpu_dcrt_poly.rs
def test_gpu_matrix_compact_cross_device_roundtrip_invariant(gpu):
    c = cpu_params(128, 2, 17, 1)
    moduli = c.to_crt()[0]
    ids = gpu.detected_gpu_device_ids()
    src_dev, dst_dev = (ids[0], ids[1]) if len(ids) >= 2 else (ids[0], ids[0])
    src = gpu.GpuDCRTPolyParams(c.ring_dimension(), moduli, c.base_bits(), gpu_ids=[src_dev])
    dst = gpu.GpuDCRTPolyParams(c.ring_dimension(), moduli, c.base_bits(), gpu_ids=[dst_dev])
    near_modulus = src.modulus() - 7
    source_eval = gpu.GpuDCRTPolyMatrix.from_poly_vec(src, [
        [const(gpu, src, 0), const(gpu, src, 1), const(gpu, src, 37)],
        [const(gpu, src, 5), gpu.GpuDCRTPoly.from_biguint_to_constant(src, near_modulus), const(gpu, src, 9)],
    ])
    source_coeff = source_eval.clone().into_coeff_domain()
    for source in (source_eval, source_coeff):
        bytes_from_src = source.to_compact_bytes()
        decoded_on_dst = gpu.GpuDCRTPolyMatrix.from_compact_bytes(dst, bytes_from_src)
        bytes_from_dst = decoded_on_dst.to_compact_bytes()
        assert bytes_from_dst == bytes_from_src
        decoded_back = gpu.GpuDCRTPolyMatrix.from_compact_bytes(src, bytes_from_dst)
        assert decoded_back == source
        assert decoded_back.to_compact_bytes() == bytes_from_src


def test_gpu_matrix_gadget_matrix(gpu, P):
    size = 3
    g = gpu.GpuDCRTPolyMatrix.gadget_matrix(P, size)
    assert g.size() == (size, size * P.modulus_bits())


def test_gpu_matrix_zero_compact_bytes_roundtrip(gpu, P):
    for nrow, ncol, level, is_ntt, max_coeff_bits in ((2, 3, 0, False, 17), (1, 4, 1, True, 23)):
        data = gpu.GpuDCRTPolyMatrix.zero_compact_bytes(P, nrow, ncol, level, is_ntt, max_coeff_bits)
        decoded = gpu.GpuDCRTPolyMatrix.from_compact_bytes(P, data)
        expected = gpu.GpuDCRTPolyMatrix._new_zero_with_state(P, nrow, ncol, level, is_ntt)
        assert decoded == expected


def _decompose_case(gpu, p, digits):
    value = 5
    row1 = [const(gpu, p, value)] + [zero(gpu, p) for _ in range(7)]
    row2 = [zero(gpu, p), const(gpu, p, value)] + [zero(gpu, p) for _ in range(6)]
    matrix = gpu.GpuDCRTPolyMatrix.from_poly_vec(p, [row1, row2])
    assert matrix.size() == (2, 8)
    gadget = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2)
    assert gadget.size() == (2, 2 * digits)
    decomposed = matrix.decompose()
    assert decomposed.size() == (2 * digits, 8)
    expected = gadget * decomposed
    assert expected.size() == (2, 8)
    assert matrix == expected


def test_gpu_matrix_decompose_basic(gpu, P):
    _decompose_case(gpu, P, P.modulus_bits())


def test_gpu_matrix_decompose_with_base8(gpu, P):
    _decompose_case(gpu, P, P.modulus_digits())


def _two_by_two(gpu, p, vals):
    return gpu.GpuDCRTPolyMatrix.from_poly_vec(p, [[const(gpu, p, vals[0]), const(gpu, p, vals[1])],
                                                    [const(gpu, p, vals[2]), const(gpu, p, vals[3])]])


def test_gpu_matrix_decompose_chunk_matches_full_decompose(gpu, P):
    matrix = _two_by_two(gpu, P, (5, 7, 11, 13))
    chunk_count = P.modulus_digits()
    full = matrix.decompose()
    chunks = [matrix.decompose_chunk(i, chunk_count) for i in range(chunk_count)]
    assert chunks[0].concat_rows(chunks[1:]) == full


def test_gpu_matrix_small_decompose_chunk_matches_full_small_decompose(gpu, P):
    matrix = _two_by_two(gpu, P, (5, 7, 11, 13))
    chunk_count = -(-P.crt_bits() // P.base_bits())
    full = matrix.small_decompose()
    chunks = [matrix.small_decompose_chunk(i, chunk_count) for i in range(chunk_count)]
    assert chunks[0].concat_rows(chunks[1:]) == full


def test_gpu_matrix_small_decompose_identity_relation(gpu, P):
    size = 3
    k = -(-P.crt_bits() // P.base_bits())
    random_int = random.randrange(0, min(P.moduli()))
    identity = gpu.GpuDCRTPolyMatrix.identity(P, size, const(gpu, P, random_int))
    decomposed = identity.small_decompose()
    assert decomposed.size() == (size * k, size)
    assert gpu.GpuDCRTPolyMatrix.small_gadget_matrix(P, size) * decomposed == identity


def test_gpu_matrix_small_decomposed_identity_chunk_digit_bound(gpu):
    p = gpu_params_from_cpu(gpu, cpu_params(128, 2, 16, 4))
    size = 4
    chunk_count = -(-p.crt_bits() // p.base_bits())
    digit_upper = 1 << p.base_bits()
    scalar = gpu.GpuDCRTPoly.from_biguint_to_constant(p, min(p.moduli()) - 1)
    for chunk_idx in range(chunk_count):
        chunk = gpu.GpuDCRTPolyMatrix.small_decomposed_identity_chunk_from_scalar(p, size, scalar, chunk_idx, chunk_count)
        assert chunk.size() == (size, size)
        for row in chunk.coeffs():
            for poly in row:
                assert all(c < digit_upper for c in poly)


def test_gpu_matrix_small_decomposed_identity_chunk_from_scalar_relation(gpu):
    p = gpu_params_from_cpu(gpu, cpu_params(128, 2, 16, 4))
    size = 4
    chunk_count = -(-p.crt_bits() // p.base_bits())
    scalar = gpu.GpuDCRTPoly.from_biguint_to_constant(p, min(p.moduli()) - 1)
    chunks = []
    for chunk_idx in range(chunk_count):
        chunk = gpu.GpuDCRTPolyMatrix.small_decomposed_identity_chunk_from_scalar(p, size, scalar, chunk_idx, chunk_count)
        assert chunk.size() == (size, size)
        chunks.append(chunk)
    from_chunks = chunks[0].concat_rows_owned(chunks[1:])
    assert from_chunks.size() == (size * chunk_count, size)
    expected_identity = gpu.GpuDCRTPolyMatrix.identity(p, size, scalar)
    assert from_chunks == expected_identity.clone().small_decompose()
    assert gpu.GpuDCRTPolyMatrix.small_gadget_matrix(p, size) * from_chunks == expected_identity


def test_gpu_matrix_mul_decompose_small_relation(gpu, P):
    n, r = 2, 2
    a = _two_by_two(gpu, P, (1, 2, 3, 4))
    assert a.size() == (r, n)
    b = _two_by_two(gpu, P, (5, 6, 7, 8))
    assert b.size() == (n, 2)
    g_small = gpu.GpuDCRTPolyMatrix.small_gadget_matrix(P, n)
    left = a.clone() * g_small
    expected = a * b
    assert left.mul_decompose_small(b) == expected


def test_gpu_matrix_gauss_samp_gq_arb_base_relation(gpu):
    p = gpu_params_from_cpu(gpu, cpu_params(128, 2, 16, 8))
    n = p.ring_dimension()
    matrix = gpu.GpuDCRTPolyMatrix.from_poly_vec(p, [[const(gpu, p, 5), const(gpu, p, 9)]])
    base = 1 << p.base_bits()
    c = (base + 1.0) * 4.578
    gadget = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, matrix.row_size())
    for offset in range(16):
        sampled = matrix.clone().gauss_samp_gq_arb_base(c, 4.578, gpu_test_seed(gpu, 0x123456789ABCDEF0, offset))
        assert gadget * sampled == matrix
    varied_poly = gpu.GpuDCRTPoly.from_coeffs(p, [(i * 7919 + 12345) & 0xFFFFFFFF for i in range(n)])
    varied = gpu.GpuDCRTPolyMatrix.from_poly_vec(p, [[varied_poly]])
    varied_gadget = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 1)
    for offset in range(16):
        sampled = varied.clone().gauss_samp_gq_arb_base(c, 4.578, gpu_test_seed(gpu, 0x00DEADBEEF, offset))
        assert varied_gadget * sampled == varied
    wide = gpu.GpuDCRTPolyMatrix.from_poly_vec(p, [[const(gpu, p, 17), const(gpu, p, 345)],
                                                   [const(gpu, p, 777), const(gpu, p, 1201)],
                                                   [const(gpu, p, 4095), const(gpu, p, 65535)]])
    wide_gadget = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, wide.row_size())
    for offset in range(16):
        sampled = wide.clone().gauss_samp_gq_arb_base(c, 4.578, gpu_test_seed(gpu, 0x55AAAA5513572468, offset))
        assert wide_gadget * sampled == wide
    rng = random.Random(2468)
    rand = gpu.GpuDCRTPolyMatrix.from_poly_vec(
        p, [[gpu.GpuDCRTPoly.from_coeffs(p, [rng.getrandbits(32) for _ in range(n)]) for _ in range(3)] for _ in range(3)])
    rand_gadget = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, rand.row_size())
    for offset in range(8):
        sampled = rand.clone().gauss_samp_gq_arb_base(c, 4.578, gpu_test_seed(gpu, 0x0F0FF0F024681357, offset))
        assert rand_gadget * sampled == rand


def test_gpu_matrix_basic_operations(gpu, P):
    zero_m = gpu.GpuDCRTPolyMatrix.zero(P, 2, 2)
    identity = gpu.GpuDCRTPolyMatrix.identity(P, 2, None)
    value = 5
    matrix1 = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[const(gpu, P, value), zero(gpu, P)], [zero(gpu, P), const(gpu, P, value)]])
    assert matrix1.entry(0, 0).coeffs()[0] == value
    matrix2 = matrix1.clone()
    assert matrix1 == matrix2
    total = matrix1.clone() + matrix2
    assert total.entry(0, 0).coeffs()[0] == 10
    assert matrix1.clone() - matrix2 == zero_m
    prod = matrix1 * identity
    assert prod.size() == (2, 2)
    assert prod.entry(0, 0).coeffs()[0] == value and prod.entry(1, 1).coeffs()[0] == value


def test_gpu_matrix_concatenation(gpu, P):
    five = gpu.GpuDCRTPoly.from_elem_to_constant(P, 5)
    matrix1 = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[five, zero(gpu, P)], [zero(gpu, P), zero(gpu, P)]])
    matrix2 = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[zero(gpu, P), zero(gpu, P)], [zero(gpu, P), five]])
    col = matrix1.concat_columns([matrix2])
    assert col.size() == (2, 4) and col.entry(0, 0).coeffs()[0] == 5 and col.entry(1, 3).coeffs()[0] == 5
    row = matrix1.concat_rows([matrix2])
    assert row.size() == (4, 2) and row.entry(0, 0).coeffs()[0] == 5 and row.entry(3, 1).coeffs()[0] == 5
    diag = matrix1.concat_diag([matrix2])
    assert diag.size() == (4, 4) and diag.entry(0, 0).coeffs()[0] == 5 and diag.entry(3, 3).coeffs()[0] == 5


def test_gpu_matrix_tensor_product(gpu, P):
    five = gpu.GpuDCRTPoly.from_elem_to_constant(P, 5)
    matrix1 = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[five, zero(gpu, P)], [zero(gpu, P), zero(gpu, P)]])
    matrix2 = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[five, zero(gpu, P)], [zero(gpu, P), zero(gpu, P)]])
    tensor = matrix1.tensor(matrix2)
    assert tensor.size() == (4, 4)
    assert tensor.entry(0, 0).coeffs()[0] == 25


def test_gpu_matrix_modulus_switch(gpu, P):
    # the four literals of gpu_dcrt_poly.rs:2675-2682 (FinRingElem::new reduces them modulo Q)
    Q = P.modulus()
    values = [1023782870921908217643761278891282178 % Q, 8179012198875468938912873783289218738 % Q,
              2034903202902173762872163465127672178 % Q, 1990091289902891278121564387120912660 % Q]
    elem = lambda v: gpu.GpuDCRTPoly.from_elem_to_constant(P, v)
    matrix = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[elem(values[0]), elem(values[1])], [elem(values[2]), elem(values[3])]])
    new_modulus = 2
    switched = matrix.modulus_switch(new_modulus)
    assert switched.params.modulus() == P.modulus()
    ms = lambda v: (v * new_modulus // Q) % new_modulus  # FinRingElem::modulus_switch, src/element/finite_ring.rs:22-26
    expected = gpu.GpuDCRTPolyMatrix.from_poly_vec(P, [[elem(ms(values[0])), elem(ms(values[1]))],
                                                       [elem(ms(values[2])), elem(ms(values[3]))]])
    assert switched == expected


# ---------------------------------------------------------------------------------------- src/poly/dcrt/gpu.rs
def test_gpu_dcrtpoly_const_coeff_u64_extracts_constant_term(gpu, P):
    for _ in range(10):
        value = random.randrange(0, 1 << 33)
        lsb_poly = gpu.GpuDCRTPoly.from_usize_to_lsb(P, value)
        poly = gpu.GpuDCRTPoly.from_usize_to_constant(P, value)
        assert poly.const_coeff_u64() == value
        assert lsb_poly.const_coeff_u64() == (value & 1)


def test_gpu_dcrtpoly_coeffs(gpu, P):
    n = P.ring_dimension()
    coeffs = [random.randrange(0, 10000) for _ in range(n)]
    assert gpu.GpuDCRTPoly.from_coeffs(P, coeffs).coeffs() == coeffs


def test_gpu_dcrtpoly_arithmetic(gpu, P):
    n = P.ring_dimension()
    coeffs1 = [100, 200, 300, 400] + [0] * (n - 4)
    coeffs2 = [500, 600, 700, 800] + [0] * (n - 4)
    poly1 = gpu.GpuDCRTPoly.from_coeffs(P, coeffs1)
    poly2 = gpu.GpuDCRTPoly.from_coeffs(P, coeffs2)
    total = poly1.clone() + poly2.clone()
    product = poly1 * poly2
    neg_poly2 = -poly2.clone()
    difference = poly1.clone() - poly2.clone()
    poly_add_assign = poly1.clone()
    poly_add_assign += poly2.clone()
    poly_mul_assign = poly1.clone()
    poly_mul_assign *= poly2.clone()
    assert total != poly1, "Sum should differ from original poly1"
    assert neg_poly2 != poly2, "Negated polynomial should differ from original"
    assert difference + poly2 == poly1, "p1 - p2 + p2 should be p1"
    assert poly_add_assign == total, "+= result should match separate +"
    assert poly_mul_assign == product, "*= result should match separate *"
    assert gpu.GpuDCRTPoly.from_usize_to_constant(P, 123) == gpu.GpuDCRTPoly.from_coeffs(P, [123] + [0] * (n - 1))
    assert gpu.GpuDCRTPoly.const_zero(P) == gpu.GpuDCRTPoly.from_coeffs(P, [0] * n)
    assert gpu.GpuDCRTPoly.const_one(P) == gpu.GpuDCRTPoly.from_coeffs(P, [1] + [0] * (n - 1))
    # the product is the negacyclic one: (100 + 200x + 300x^2 + 400x^3)(500 + 600x + 700x^2 + 800x^3)
    want = [0] * n
    for i, a in enumerate(coeffs1[:4]):
        for j, b in enumerate(coeffs2[:4]):
            want[i + j] += a * b
    assert product.coeffs() == want


def _sampled_poly(gpu, p, dist):
    return gpu.GpuDCRTPolyUniformSampler().sample_poly(p, dist)


def test_gpu_dcrtpoly_partial_eq_across_domains(gpu, P):
    eval_poly = _sampled_poly(gpu, P, gpu.DistType.FinRingDist())
    coeff_poly = eval_poly.ensure_coeff_domain()
    assert not coeff_poly.is_ntt() and eval_poly.is_ntt()
    assert coeff_poly == eval_poly, "PartialEq should match across coeff/eval domains"


def test_gpu_dcrtpoly_decompose(gpu, P):
    poly = _sampled_poly(gpu, P, gpu.DistType.FinRingDist())
    assert len(poly.decompose_base()) == P.modulus_digits()


def test_gpu_dcrtpoly_to_compact_bytes_bit_dist(gpu, P):
    poly = _sampled_poly(gpu, P, gpu.DistType.BitDist())
    data = poly.to_compact_bytes()
    assert data, "compact serialization should not be empty"
    assert gpu.GpuDCRTPoly.from_compact_bytes(P, data) == poly


def test_gpu_dcrtpoly_to_compact_bytes_uniform_dist(gpu, P):
    poly = _sampled_poly(gpu, P, gpu.DistType.FinRingDist())
    data = poly.to_compact_bytes()
    assert data, "compact serialization should not be empty"
    assert gpu.GpuDCRTPoly.from_compact_bytes(P, data) == poly


def test_gpu_dcrtpoly_from_compact_bytes(gpu, P):
    for dist in (gpu.DistType.BitDist(), gpu.DistType.FinRingDist(), gpu.DistType.GaussDist(3.2), gpu.DistType.TernaryDist()):
        original = _sampled_poly(gpu, P, dist)
        assert gpu.GpuDCRTPoly.from_compact_bytes(P, original.to_compact_bytes()) == original


# ------------------------------------------------------------------------------------------- src/sampler/gpu.rs
def test_gpu_uniform_sampler_size(gpu, PS):
    sampled = gpu.GpuDCRTPolyUniformSampler().sample_uniform(PS, 3, 4, gpu.DistType.FinRingDist())
    assert sampled.row_size() == 3 and sampled.col_size() == 4


def test_gpu_hash_sampler_is_deterministic(gpu, PS):
    sampler = gpu.GpuDCRTPolyHashSampler("keccak256")
    key, tag = bytes([7] * 32), b"gpu-hash"
    assert sampler.sample_hash(PS, key, tag, 4, 5, gpu.DistType.FinRingDist()) == sampler.sample_hash(PS, key, tag, 4, 5, gpu.DistType.FinRingDist())


def test_gpu_hash_sampler_decomposed_matches_legacy_path(gpu, PS):
    sampler = gpu.GpuDCRTPolyHashSampler("keccak256")
    key, tag = bytes([11] * 32), b"gpu-hash-decomposed"
    decomposed = sampler.sample_hash_decomposed(PS, key, tag, 3, 4, gpu.DistType.FinRingDist())
    legacy = sampler.sample_hash(PS, key, tag, 3, 4, gpu.DistType.FinRingDist())
    assert decomposed == legacy.decompose()


def test_gpu_hash_sampler_column_subrange_matches_full_sample(gpu, PS):
    sampler = gpu.GpuDCRTPolyHashSampler("keccak256")
    key, tag = bytes([13] * 32), b"gpu-hash-columns"
    d = gpu.DistType.FinRingDist()
    full = sampler.sample_hash(PS, key, tag, 4, 9, d)
    assert sampler.sample_hash_columns(PS, key, tag, 4, 9, 2, 3, d) == full.slice_columns(2, 5)
    assert sampler.sample_hash_decomposed_columns(PS, key, tag, 4, 9, 2, 3, d) == full.slice_columns(2, 5).decompose()
    assert sampler.sample_hash_small_decomposed_columns(PS, key, tag, 4, 9, 2, 3, d) == full.slice_columns(2, 5).small_decompose()


def test_sample_gpu_matrix_with_seed_gauss_coeff_lt_6sigma(gpu, PS):
    from mxx_amd.sampler import sample_gpu_matrix_with_seed

    sigma = 4.578
    sampled = sample_gpu_matrix_with_seed(PS, 4, 5, gpu.DistType.GaussDist(sigma), gpu.GpuRngSeed.from_bytes(bytes([0x5A] * 32)))
    strict_upper = math.ceil(sigma * 6.0)
    q = PS.modulus()
    for row in sampled.coeffs():
        for poly in row:
            for value in poly:
                assert min(value, q - value) < strict_upper


# ---------------------------------------------------------------------------------- src/sampler/trapdoor/gpu.rs
def test_gpu_trapdoor_generation(gpu, PS):
    size = 3
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(PS, SIGMA)
    trapdoor, public_matrix = sampler.trapdoor(PS, size)
    assert public_matrix.row_size() == size
    assert public_matrix.col_size() == (PS.modulus_digits() + 2) * size
    k = PS.modulus_digits()
    identity = gpu.GpuDCRTPolyMatrix.identity(PS, size * k, None)
    trapdoor_matrix = trapdoor.r.concat_rows([trapdoor.e, identity])
    assert public_matrix * trapdoor_matrix == gpu.GpuDCRTPolyMatrix.gadget_matrix(PS, size)


def test_gpu_trapdoor_round_trip_bytes(gpu, PS):
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(PS, SIGMA)
    trapdoor, _public = sampler.trapdoor(PS, 3)
    data = gpu.GpuDCRTPolyTrapdoorSampler.trapdoor_to_bytes(trapdoor)
    decoded = gpu.GpuDCRTPolyTrapdoorSampler.trapdoor_from_bytes(PS, data)
    assert decoded is not None, "trapdoor bytes should decode"
    assert gpu.GpuDCRTPolyTrapdoorSampler.trapdoor_to_bytes(decoded) == data


def test_gpu_preimage_generation_square(gpu, PS):
    size = 3
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(PS, SIGMA)
    trapdoor, public_matrix = sampler.trapdoor(PS, size)
    target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(PS, size, size, gpu.DistType.FinRingDist())
    preimage = sampler.preimage(PS, trapdoor, public_matrix, target)
    assert public_matrix * preimage == target


def test_gpu_preimage_reuses_trapdoor_cache_for_distinct_targets(gpu, PS):
    size = 3
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(PS, SIGMA)
    trapdoor, public_matrix = sampler.trapdoor(PS, size)
    us = gpu.GpuDCRTPolyUniformSampler()
    first = us.sample_uniform(PS, size, size, gpu.DistType.FinRingDist())
    second = us.sample_uniform(PS, size, size, gpu.DistType.FinRingDist())
    assert first != second, "targets should differ"
    x1 = sampler.preimage(PS, trapdoor, public_matrix, first)
    cache = trapdoor._p1_cache
    x2 = sampler.preimage(PS, trapdoor, public_matrix, second)
    assert trapdoor._p1_cache is cache, "the covariance cache is built once per trapdoor"
    assert public_matrix * x1 == first and public_matrix * x2 == second


def test_gpu_preimage_generation_square_not_plain_gadget_solution(gpu, PS):
    size = 3
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(PS, SIGMA)
    trapdoor, public_matrix = sampler.trapdoor(PS, size)
    target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(PS, size, size, gpu.DistType.FinRingDist())
    z_plain = target.decompose()
    z_plain_full = (trapdoor.r * z_plain).concat_rows([trapdoor.e * z_plain]).concat_rows([z_plain])
    assert public_matrix * z_plain_full == target
    sampled = sampler.preimage(PS, trapdoor, public_matrix, target)
    assert public_matrix * sampled == target
    assert sampled != z_plain_full, "preimage sampler should not collapse to the plain deterministic gadget preimage"


def test_gpu_preimage_sampler_parameters_follow_instance_sigma(gpu):
    from mxx_amd.trapdoor import p1_covariance_parameters, preimage_c, preimage_smoothing_parameter

    p = gpu_params_from_cpu(gpu, cpu_params(1 << 10, 5, 51, 17))
    base = 1 << p.base_bits()
    default_sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, SIGMA)
    larger_sigma = SIGMA * 1.5
    larger_sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, larger_sigma)
    n, k, size = p.ring_dimension(), p.modulus_digits(), 2
    default_s = preimage_smoothing_parameter(base, default_sampler.sigma, size, n, k)
    larger_s = preimage_smoothing_parameter(base, larger_sampler.sigma, size, n, k)
    assert default_sampler.c == preimage_c(base, SIGMA)
    assert larger_sampler.c == preimage_c(base, larger_sigma)
    assert p1_covariance_parameters(p, size, larger_sigma) == (larger_sampler.c, larger_s, larger_sigma)
    assert larger_sampler.c > default_sampler.c and larger_s > default_s


def _norm_bound(p, size, bound_sigma):
    from mxx_amd.trapdoor import compute_preimage_norm

    return compute_preimage_norm(math.sqrt(p.ring_dimension()), size * p.modulus_digits(), float(1 << p.base_bits()), None, bound_sigma)


def _max_centred(matrix):
    q = matrix.params.modulus()
    return max(min(v, q - v) for row in matrix.coeffs() for poly in row for v in poly)


def _assert_preimage_reconstructs_target_and_respects_norm_bound(gpu, sigma, bound_sigma):
    size = 2
    p = gpu_params_from_cpu(gpu, cpu_params(1 << 10, 5, 51, 17))
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, sigma)
    trapdoor, public_matrix = sampler.trapdoor(p, size)
    us = gpu.GpuDCRTPolyUniformSampler()
    bound = _norm_bound(p, size, bound_sigma)
    for _sample in range(4):
        target = us.sample_uniform(p, size, size, gpu.DistType.FinRingDist())
        preimage = sampler.preimage(p, trapdoor, public_matrix, target)
        assert public_matrix * preimage == target
        assert _max_centred(preimage) < bound


def test_gpu_preimage_coefficients_below_compute_preimage_norm(gpu):
    _assert_preimage_reconstructs_target_and_respects_norm_bound(gpu, SIGMA, None)


def test_gpu_preimage_coefficients_below_compute_preimage_norm_non_default_sigma(gpu):
    sigma = SIGMA * 1.25
    _assert_preimage_reconstructs_target_and_respects_norm_bound(gpu, sigma, sigma)


def test_gpu_p_hat_coefficients_below_compute_preimage_norm(gpu):
    from mxx_amd.trapdoor import preimage_c, preimage_smoothing_parameter

    size = 2
    p = gpu_params_from_cpu(gpu, cpu_params(1 << 10, 5, 51, 17))
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, SIGMA)
    trapdoor, _public = sampler.trapdoor(p, size)
    bound = _norm_bound(p, size, None)
    n, k, base = p.ring_dimension(), p.modulus_digits(), 1 << p.base_bits()
    c = preimage_c(base, SIGMA)
    s = preimage_smoothing_parameter(base, SIGMA, size, n, k)
    dgg_large_std = math.sqrt(s * s - c * c)
    for _sample in range(4):
        p_hat = sampler.sample_pert_square_mat_gpu_native(p, trapdoor, s, c, SIGMA, dgg_large_std, size)  # :502-541
        assert p_hat.size() == (2 * size + size * p.modulus_digits(), size)
        assert _max_centred(p_hat) < bound


def test_gpu_preimage_compact_cross_device_restore_relation_and_norm(gpu):
    size = 2
    base_params = gpu_params_from_cpu(gpu, cpu_params(1 << 10, 5, 51, 17))
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(base_params, SIGMA)
    us = gpu.GpuDCRTPolyUniformSampler()
    bound = _norm_bound(base_params, size, None)
    ids = gpu.detected_gpu_device_ids()
    devices = ids if len(ids) >= 2 else [ids[0], ids[0]]  # one GPU: two contexts on it
    cases = []
    for idx, src_device in enumerate(devices):
        dst_device = devices[(idx + 1) % len(devices)]
        src_params = gpu.GpuDCRTPolyParams(base_params.ring_dimension(), base_params.moduli(), base_params.base_bits(), gpu_ids=[src_device])
        trapdoor, public_matrix = sampler.trapdoor(src_params, size)
        target = us.sample_uniform(src_params, size, size, gpu.DistType.FinRingDist())
        preimage = sampler.preimage(src_params, trapdoor, public_matrix, target)
        assert public_matrix * preimage == target
        cases.append((dst_device, public_matrix.to_compact_bytes(), target.to_compact_bytes(), preimage.to_compact_bytes()))
    for dst_device, pub_bytes, target_bytes, preimage_bytes in cases:
        dst_params = gpu.GpuDCRTPolyParams(base_params.ring_dimension(), base_params.moduli(), base_params.base_bits(), gpu_ids=[dst_device])
        public_matrix = gpu.GpuDCRTPolyMatrix.from_compact_bytes(dst_params, pub_bytes)
        target = gpu.GpuDCRTPolyMatrix.from_compact_bytes(dst_params, target_bytes)
        preimage = gpu.GpuDCRTPolyMatrix.from_compact_bytes(dst_params, preimage_bytes)
        assert public_matrix * preimage == target
        assert _max_centred(preimage) < bound
