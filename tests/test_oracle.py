"""CPU-only: pins the oracle (oracle/oracle.c) by convention-independent checks.

The reference holds no golden vectors for this path; what it does
hold are algebraic predicates and the OpenFHE-convention restatement in
src/gadgets/ntt/mod.rs.  These tests check the oracle against both, against the
CRT bases SURVEY.md Appendix A.1 lists, and against the committed fixtures.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

SURVEY_A1 = {
    (16384, 15, 24): [16580609, 16515073, 16384001, 16121857, 15630337, 14942209, 14909441, 14155777, 14123009,
                      13664257, 13631489, 13565953, 13336577, 13238273, 13074433],
    (4, 2, 17): [131041, 131009],
    (128, 2, 17): [130817, 129793],
    (128, 2, 16): [64513, 61441],
    (256, 3, 51): [2251799813684737, 2251799813667841, 2251799813640193],
    (1024, 5, 51): [2251799813640193, 2251799813632001, 2251799813613569, 2251799813560321, 2251799813554177],
}


@pytest.mark.parametrize("key", list(SURVEY_A1))
def test_crt_basis_matches_survey(key):
    assert O.gen_crt_basis(*key) == SURVEY_A1[key]


def test_crt_basis_reference_held_datum():
    """The one concrete value on this path that the reference itself holds: its GPU slot-transfer test builds
    `GpuDCRTPolyParams::new(4, vec![131041, 131009], 1)` (src/slot_transfer/bgg_poly_encoding_gpu.rs:1501) - the
    moduli OpenFHE's GenCRTBasis returns for DCRTPolyParams::default() = (n=4, depth 2, 17 bits),
    src/poly/dcrt/params.rs:60-74.  The basis rule restated here (largest `bits`-bit prime = 1 mod 2n, then the
    next smaller ones) reproduces it; every other entry of SURVEY_A1 is derived by the same rule and is NOT
    pinned by the reference (parity stays unpinned at the byte level)."""
    assert O.gen_crt_basis(4, 2, 17) == [131041, 131009]
    from mxx_amd.params import DCRTPolyParams

    assert DCRTPolyParams().to_crt()[0] == [131041, 131009]


def test_reference_params_tests_replayed():
    """Every case of the reference's own `DCRTPolyParams` tests (src/poly/dcrt/params.rs:116-213), replayed on the
    host mirror and on the oracle's basis rule: ring dimensions 16 / 2 / 1 with modulus_bits == 204, modulus_bits ==
    depth * bits for depths 4..6 at 51 bits and depth 7 at 20 bits, base_bits 1 / 4 / 20 kept, and the panic message
    for a ring dimension that is not a power of two.  These pin the bit length of Q and the basis rule's divisibility
    (q = 1 mod 2n), not the prime values themselves."""
    from mxx_amd.params import DCRTPolyParams

    for n in (16, 2, 1):
        p = DCRTPolyParams(n, 4, 51, 1)
        assert p.ring_dimension() == n and p.modulus_bits() == 204 and p.base_bits() == 1
    for depth, bits in ((4, 51), (5, 51), (6, 51), (7, 20)):
        p = DCRTPolyParams(16, depth, bits, 1)
        assert p.ring_dimension() == 16 and p.modulus_bits() == depth * bits
        basis = O.gen_crt_basis(16, depth, bits)
        Q = 1
        for q in basis:
            Q *= q
            assert (q - 1) % 32 == 0
        assert Q.bit_length() == depth * bits and p.to_crt()[0] == basis
    for base in (1, 4, 20):
        assert DCRTPolyParams(16, 4, 51, base).base_bits() == base
    with pytest.raises(ValueError, match="ring_dimension must be a power of 2"):
        DCRTPolyParams(20, 4, 51, 1)


@pytest.mark.parametrize("n,bits", [(4, 17), (16, 18), (128, 17), (128, 16), (256, 51), (1024, 51), (4096, 24)])
def test_ntt_convention_and_roundtrip(n, bits):
    rng = np.random.default_rng(n * 131 + bits)
    for q in O.gen_crt_basis(n, 2, bits):
        psi = O.min_primitive_root(q, 2 * n)
        assert pow(psi, n, q) == q - 1  # primitive 2n-th root
        # minimality over all odd powers
        if n <= 256:
            assert psi == min(pow(psi, 2 * j + 1, q) for j in range(n))
        a = rng.integers(0, q, n, dtype=np.uint64)
        A = O.ntt_vec(a, q)
        assert np.array_equal(A, O.ntt_vec(a, q, plain=True))  # two implementations agree
        assert np.array_equal(O.ntt_vec(A, q, inverse=True), a)
        assert np.array_equal(O.ntt_vec(A, q, inverse=True, plain=True), a)
        # slot k holds a(psi^(2*bitrev(k)+1))  
        lg = n.bit_length() - 1
        for k in range(min(n, 6)):
            br = int(format(k, f"0{lg}b")[::-1], 2) if lg else 0
            x = pow(psi, 2 * br + 1, q)
            acc = 0
            for i in range(n - 1, -1, -1):
                acc = (acc * x + int(a[i])) % q
            assert acc == int(A[k])


@pytest.mark.parametrize("n,bits", [(4, 17), (16, 18), (128, 17), (256, 51)])
def test_ring_product_vs_schoolbook(n, bits):
    rng = np.random.default_rng(7 + n)
    moduli = O.gen_crt_basis(n, 3, bits)
    a = O.random_matrix(11, 1, 2, moduli, n)
    b = O.random_matrix(12, 1, 2, moduli, n)
    c = O.ring_mul_batch(a, b, moduli)
    for p in range(2):
        for l, q in enumerate(moduli):
            assert np.array_equal(c[0, p, l], O.negacyclic_schoolbook(a[0, p, l], b[0, p, l], q))


def test_matmul_fast_equals_plain():
    n = 64
    moduli = O.gen_crt_basis(n, 3, 30)
    a = O.random_matrix(1, 3, 5, moduli, n)
    b = O.random_matrix(2, 5, 4, moduli, n)
    assert np.array_equal(O.matmul(a, b, moduli), O.matmul(a, b, moduli, fast=True))


@pytest.mark.parametrize("n,depth,bits,base", [(16, 2, 17, 1), (16, 2, 16, 4), (16, 2, 16, 8), (16, 3, 17, 5), (8, 2, 51, 17)])
def test_gadget_times_decompose_is_identity(n, depth, bits, base):
    """G * G^-1(M) == M  (reference: src/matrix/dcrt_poly.rs:514-601,682-728)."""
    moduli = O.gen_crt_basis(n, depth, bits)
    M = O.random_matrix(5, 2, 3, moduli, n)  # coefficient domain
    dec = O.decompose(M, moduli, base)
    G = O.gadget_matrix(2, moduli, n, base)  # EVAL
    prod = O.matmul(G, O.matrix_ntt(dec, moduli), moduli)
    assert np.array_equal(O.matrix_ntt(prod, moduli, inverse=True), M)
    # digits are small: below 2^base in every limb
    assert int(dec.max()) < (1 << base)


def test_small_gadget_times_small_decompose():
    """G_small * small_decompose(M) == M when ||M||inf < min q_i (dcrt_poly.rs:603-680)."""
    n, base = 16, 4
    moduli = O.gen_crt_basis(n, 2, 16)
    small = (O.splitmix64(3, 2 * 2 * n).reshape(2, 2, 1, n) % np.uint64(min(moduli))).astype(np.uint64)
    M = np.repeat(small, len(moduli), axis=2)  # same small integer in every limb
    dec = O.decompose(M, moduli, base, small=True)
    G = O.gadget_matrix(2, moduli, n, base, small=True)
    prod = O.matmul(G, O.matrix_ntt(dec, moduli), moduli)
    assert np.array_equal(O.matrix_ntt(prod, moduli, inverse=True), M)


def test_crt_reconstruct():
    moduli = O.gen_crt_basis(16, 3, 20)
    Q = moduli[0] * moduli[1] * moduli[2]
    x = 123456789012345 % Q
    assert O.crt_reconstruct([x % q for q in moduli], moduli) == x


def test_golden_fixtures_match_oracle():
    """Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    files = sorted(f for f in os.listdir(gdir) if f.endswith(".npz") and not f.startswith("samplers_"))
    assert files, "no golden fixtures committed"
    for f in files:
        z = np.load(os.path.join(gdir, f))
        moduli = [int(q) for q in z["moduli"]]
        assert np.array_equal(O.matrix_ntt(z["a_coeff"], moduli), z["a_eval"])
        assert np.array_equal(O.matmul(z["a_eval"], z["b_eval"], moduli), z["ab_eval"])
        assert np.array_equal(O.decompose(z["m_coeff"], moduli, int(z["base_bits"])), z["m_decomposed"])
        # schoolbook pin of one ring product inside the fixture
        q0 = moduli[0]
        a0 = z["a_coeff"][0, 0, 0]
        b0 = O.matrix_ntt(z["b_eval"], moduli, inverse=True)[0, 0, 0]
        c0 = O.negacyclic_schoolbook(a0, b0, q0)
        prod = O.pointwise("mul", z["a_eval"][:1, :1], z["b_eval"][:1, :1], moduli)
        assert np.array_equal(O.matrix_ntt(prod, moduli, inverse=True)[0, 0, 0], c0)


def test_golden_sampler_fixtures_match_oracle():
    """tests/golden/samplers_*.npz pin the seeded samplers' keying (round 3) as data: every distribution and its column
    window, 18-bit and 51-bit limbs."""
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    files = sorted(f for f in os.listdir(gdir) if f.startswith("samplers_"))
    assert len(files) >= 4
    for f in files:
        z = np.load(os.path.join(gdir, f))
        moduli, n, seed = [int(q) for q in z["moduli"]], int(z["n"]), bytes(z["seed"])
        # samplers_refkey_*: the reference device RNG's own keying (MXX_HIP_RNG_COMPAT=reference), round 4
        sample = O.sample_distribution_refkey if f.startswith("samplers_refkey_") else O.sample_distribution
        for dist, sigma in (("uniform", 0.0), ("bit", 0.0), ("ternary", 0.0), ("gauss", 4.578)):
            assert np.array_equal(sample(2, 3, moduli, n, dist, sigma, seed), z[dist]), (f, dist)
            assert np.array_equal(z[dist + "_window"], z[dist][:, 1:3]), (f, dist)
