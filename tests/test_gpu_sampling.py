"""GPU parity for the seeded samplers, the G-lattice sampler, the p1 sampler and preimages.

Integer-only samplers (uniform / bit / ternary / Karney Gaussian, p1 conditional
sampling) are compared BIT-EXACTLY with the CPU restatement (same ChaCha streams, IEEE
double arithmetic without contraction).  The G-lattice sampler uses log/cos on its
Box-Muller perturbation, which differ in the last ulp between libm and the device, so
it is held to the reference's own acceptance predicates instead
(src/matrix/gpu_dcrt_poly.rs:2381-2542, src/sampler/trapdoor/gpu.rs:547-811).
"""
import math

import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu

SEED_BYTES = bytes((7 * i + 3) & 0xFF for i in range(32))


def seed(gpu, salt=0):
    b = bytearray(SEED_BYTES)
    b[0] ^= salt
    return gpu.GpuRngSeed.from_bytes(bytes(b))


def test_sampler_golden_fixtures_on_gpu(gpu, oracle, hip_env):
    """The committed sampler fixtures (tests/golden/samplers_*.npz) against the device: every distribution, the full
    matrix and its column window."""
    import os

    gdir = os.path.join(os.path.dirname(__file__), "golden")
    for f in sorted(x for x in os.listdir(gdir) if x.startswith("samplers_")):
        # samplers_refkey_*: the same calls under MXX_HIP_RNG_COMPAT=reference (the reference device RNG's own keying)
        if f.startswith("samplers_refkey_"):
            hip_env.set("MXX_HIP_RNG_COMPAT", "reference")
        else:
            hip_env.unset("MXX_HIP_RNG_COMPAT")
        z = np.load(os.path.join(gdir, f))
        moduli, n = [int(q) for q in z["moduli"]], int(z["n"])
        p = gpu.GpuDCRTPolyParams(n, moduli, 6)
        s = gpu.GpuRngSeed.from_bytes(bytes(z["seed"]))
        for dist, sigma in (("uniform", 0.0), ("bit", 0.0), ("ternary", 0.0), ("gauss", 4.578)):
            code = oracle.DIST[dist]
            m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, sigma, s)
            assert np.array_equal(m.to_coeff_rns(), z[dist]), (f, dist)
            w = gpu.GpuDCRTPolyMatrix.sample_distribution_columns(p, 2, 3, 1, 2, code, sigma, s)
            assert np.array_equal(w.to_coeff_rns(), z[dist + "_window"]), (f, dist)


def test_uniform_sampler_overflow_streams_bit_exact(gpu, oracle):
    """Moduli that reject one 64-bit draw in nine: the rare path of the block-per-eight-draws uniform sampler (the
    coefficient's overflow stream) runs hundreds of times and must agree with the CPU restatement; windows commute."""
    from conftest import high_rejection_moduli

    n = 256
    moduli = high_rejection_moduli(n, 3)
    p = gpu.GpuDCRTPolyParams(n, moduli, 20)
    s = seed(gpu, 9)
    code = oracle.DIST["uniform"]
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, 0.0, s)
    want = oracle.sample_distribution(2, 3, moduli, n, "uniform", 0.0, s)
    assert np.array_equal(m.to_coeff_rns(), want)
    w = gpu.GpuDCRTPolyMatrix.sample_distribution_columns(p, 2, 3, 2, 1, code, 0.0, s)
    assert np.array_equal(w.to_coeff_rns(), want[:, 2:3])


@pytest.mark.parametrize("n,depth,bits,base", [(4, 2, 17, 1), (16, 3, 18, 6), (128, 2, 17, 1), (1024, 2, 24, 12), (64, 2, 51, 17),
                                               (16384, 2, 24, 12)])
@pytest.mark.parametrize("dist,sigma", [("uniform", 0.0), ("gauss", 4.578), ("gauss", 321.7), ("bit", 0.0), ("ternary", 0.0)])
def test_sample_distribution_bit_exact(gpu, oracle, n, depth, bits, base, dist, sigma):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    s = seed(gpu)
    code = oracle.DIST[dist]
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, sigma, s)
    assert m.is_ntt  # always EVAL (MatrixSampling.cu:463-469)
    want = oracle.sample_distribution(2, 3, moduli, n, dist, sigma, s)
    assert np.array_equal(m.to_coeff_rns(), want)
    assert np.array_equal(m.to_rns(), oracle.matrix_ntt(want, moduli))
    # column window == slice of the full sample (src/sampler/gpu.rs:323-361)
    w = gpu.GpuDCRTPolyMatrix.sample_distribution_columns(p, 2, 3, 1, 2, code, sigma, s)
    assert w == m.slice(0, 2, 1, 3)
    # a different seed gives a different matrix
    assert not (gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, sigma, seed(gpu, 1)) == m)


@pytest.mark.parametrize("sigma", [0.35, 1.0, 2.5, 17.25, 1.0e5, 3.0e7])
def test_gauss_widths_from_below_one_to_above_the_modulus(gpu, oracle, sigma):
    """Karney's machine at the edges of its parameter range: ceil(sigma) = 1 (the offset draw is always 0), widths
    around 1, and widths whose samples wrap around the 24-bit moduli many times (residues of large signed integers)."""
    p = make_params(gpu, oracle, 256, 2, 24, 12)
    s = seed(gpu, 41)
    want = oracle.sample_distribution(3, 2, p.moduli(), 256, "gauss", sigma, s)
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 3, 2, oracle.DIST["gauss"], sigma, s)
    assert np.array_equal(m.to_coeff_rns(), want)


def test_gauss_million_samples_cover_the_tie_path(gpu, oracle):
    """2^20 Gaussian integers, bit for bit: with 16-bit draws about one comparison in 65536 ties and takes the
    low-bits path (KS_TIE on the device, lz_less in the restatement) - a few hundred times here, counted by the oracle."""
    p = make_params(gpu, oracle, 16384, 2, 24, 12)
    s = seed(gpu, 33)
    before = oracle.karney_ties()
    want = oracle.sample_distribution(8, 8, p.moduli(), 16384, "gauss", 4.578, s)
    ties = oracle.karney_ties() - before
    assert ties >= 100, ties
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 8, 8, oracle.DIST["gauss"], 4.578, s)
    assert np.array_equal(m.to_coeff_rns(), want)


@pytest.mark.parametrize("per_lane", ["1", "3", "16"])
@pytest.mark.parametrize("n,rows,cols,sigma", [(128, 2, 3, 4.578), (1024, 1, 5, 8191.5)])
def test_gauss_persistent_lanes_bit_exact(gpu, oracle, hip_env, per_lane, n, rows, cols, sigma):
    """Lanes that work through several coefficients (stream switch, parked states, idle tails)
    still produce the sequential sampler's integers."""
    hip_env.set("MXX_HIP_SAMPLER_PER_LANE", per_lane)
    p = make_params(gpu, oracle, n, 2, 24, 12)
    s = seed(gpu, 9)
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, rows, cols, oracle.DIST["gauss"], sigma, s)
    want = oracle.sample_distribution(rows, cols, p.moduli(), n, "gauss", sigma, s)
    assert np.array_equal(m.to_coeff_rns(), want)


@pytest.mark.parametrize("fill_every,per_lane", [("1", "2"), ("5", "7"), ("8", "3")])
def test_samples_do_not_depend_on_the_refill_cadence(gpu, oracle, hip_env, fill_every, per_lane):
    """The lane kernels' keystream refills are scheduled every k-th checkpoint and a lane that runs dry waits; a sample
    is a function of its stream alone, so the Gaussian matrix, the G-sampler's digits and a whole preimage stay
    bit-identical to the CPU restatement under any cadence and any number of elements per lane."""
    from mxx_amd.sampler import seed_source

    hip_env.set("MXX_HIP_SAMPLER_FILL_EVERY", fill_every)
    hip_env.set("MXX_HIP_SAMPLER_PER_LANE", per_lane)
    n, depth, bits, base = 1024, 2, 24, 12
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    s = seed(gpu, 12)
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, oracle.DIST["gauss"], 321.7, s)
    assert np.array_equal(m.to_coeff_rns(), oracle.sample_distribution(2, 3, moduli, n, "gauss", 321.7, s))
    M = rand_matrix(oracle, 91, 2, 2, moduli, n)
    c = ((1 << base) + 1) * 4.578
    z = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False).gauss_samp_gq_arb_base(c, 4.578, s).to_coeff_rns()
    assert np.array_equal(z, oracle.gauss_samp_gq(M, moduli, base, c, s))
    master = bytes(range(32))
    seeds = [oracle._seed_from(master, i).tobytes() for i in range(6)]
    r, e, a = oracle.trapdoor_gen(moduli, n, base, 4.578, 1, master)
    target = oracle.matrix_ntt(rand_matrix(oracle, 78, 1, 2, moduli, n), moduli)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    with seed_source(seeds):
        td, A = sampler.trapdoor(p, 1)
        x = sampler.preimage(p, td, A, gpu.GpuDCRTPolyMatrix.from_rns(p, target, True))
    assert np.array_equal(x.ensure_eval().to_rns(), oracle.preimage(moduli, n, base, 4.578, r, e, a, target, master))


@pytest.mark.parametrize("per_lane", ["1", "5"])
@pytest.mark.parametrize("n,depth,bits,base", [(1024, 2, 24, 12), (128, 2, 17, 6), (64, 2, 51, 17), (128, 2, 16, 16)])
def test_gauss_samp_gq_lane_form_equals_simple_form(gpu, oracle, hip_env, per_lane, n, depth, bits, base):
    """The two-pass persistent-lane G-sampler and the one-thread-per-element kernel consume the
    same streams in the same order: identical digits (dpt = 2, 3, 3, 1)."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    M = rand_matrix(oracle, 123, 2, 3, p.moduli(), n)
    c = ((1 << base) + 1) * 4.578
    s = seed(gpu, 11)
    hip_env.set("MXX_HIP_SAMPLER_PER_LANE", per_lane)
    hip_env.unset("MXX_HIP_GSAMP")
    z_lanes = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False).gauss_samp_gq_arb_base(c, 4.578, s)
    hip_env.set("MXX_HIP_GSAMP", "simple")
    z_simple = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False).gauss_samp_gq_arb_base(c, 4.578, s)
    assert z_lanes == z_simple
    if bits > base:  # one digit per tower: the reference's formula (MatrixTrapdoor.cu:811-814) adds base*z, so
        # G*z == M is not expected there; the forms must still agree bit for bit
        assert gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2) * z_lanes == gpu.GpuDCRTPolyMatrix.from_rns(p, M, False).ensure_eval()


def test_hash_sampler_determinism_and_uniform_sampler(gpu, oracle):
    p = make_params(gpu, oracle, 128, 2, 17, 1)
    hs = gpu.GpuDCRTPolyHashSampler()
    key = bytes(range(32))
    a = hs.sample_hash(p, key, b"tag", 2, 4, gpu.DistType.FinRingDist())
    b = hs.sample_hash(p, key, b"tag", 2, 4, gpu.DistType.FinRingDist())
    c = hs.sample_hash(p, key, b"other", 2, 4, gpu.DistType.FinRingDist())
    assert a == b and not (a == c)
    cols = hs.sample_hash_columns(p, key, b"tag", 2, 4, 1, 2, gpu.DistType.FinRingDist())
    assert cols == a.slice(0, 2, 1, 3)
    dec = hs.sample_hash_decomposed(p, key, b"tag", 2, 4, gpu.DistType.FinRingDist())
    assert gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2) * dec == a
    us = gpu.GpuDCRTPolyUniformSampler()
    g = us.sample_uniform(p, 3, 3, gpu.DistType.GaussDist(3.2))
    c0 = oracle.centered(g.to_coeff_rns()[:, :, 0], p.moduli()[0])
    assert np.abs(c0).max() < 6 * 3.2  # |x| < 6 sigma (src/sampler/gpu.rs:363-400)
    assert us.sample_poly(p, gpu.DistType.BitDist()).inner.size() == (1, 1)


@pytest.mark.parametrize("n,depth,bits,base", [(128, 2, 17, 1), (128, 2, 16, 4), (128, 2, 16, 8), (16, 3, 17, 5), (64, 2, 51, 17), (1024, 2, 24, 12)])
def test_gauss_samp_gq_relation(gpu, oracle, n, depth, bits, base):
    """G * gauss_samp_gq_arb_base(M) == M over several seeds (gpu_dcrt_poly.rs:2381-2542)."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    G = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2)
    c = ((1 << base) + 1) * 4.578
    for salt in range(4):
        M = rand_matrix(oracle, 80 + salt, 2, 3, moduli, n)
        gm = gpu.GpuDCRTPolyMatrix.from_rns(p, M, True if salt % 2 else False)
        keep = gm.clone().ensure_eval()
        z = gm.gauss_samp_gq_arb_base(c, 4.578, seed(gpu, salt))
        assert z.is_ntt and z.size() == (2 * p.modulus_digits(), 3)
        assert G * z == keep
        zc = oracle.centered(z.to_coeff_rns()[:, :, 0], moduli[0])
        assert np.abs(zc).max() < 8 * c
        # digits are consistent across limbs (one integer per coefficient)
        zr = z.to_coeff_rns()
        for l, q in enumerate(moduli):
            assert np.array_equal(oracle.centered(zr[:, :, l], q), zc)


@pytest.mark.parametrize("n,depth,bits,base,form", [
    (1024, 2, 24, 12, "lanes"), (1024, 2, 24, 12, "simple"), (128, 2, 17, 6, "lanes"), (64, 2, 51, 17, "lanes"),
    (128, 2, 16, 2, "simple"),  # dpt = 8: the one-thread-per-element kernel
])
def test_gauss_samp_gq_bit_exact(gpu, oracle, hip_env, n, depth, bits, base, form):
    """Same algorithm, same streams, and Box-Muller's log / cos are fixed IEEE operation sequences
    compiled into both sides (mxx_amd/csrc/detmath.h): every digit equals the CPU restatement's."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    M = rand_matrix(oracle, 90, 2, 2, moduli, n)
    c = ((1 << base) + 1) * 4.578
    s = seed(gpu, 5)
    if form == "simple":
        hip_env.set("MXX_HIP_GSAMP", "simple")
    z_gpu = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False).gauss_samp_gq_arb_base(c, 4.578, s).to_coeff_rns()
    assert np.array_equal(z_gpu, oracle.gauss_samp_gq(M, moduli, base, c, s))


@pytest.mark.parametrize("assembly", ["small", "large"])
@pytest.mark.parametrize("n,depth,bits,base,d,cols", [(256, 2, 24, 12, 1, 3), (64, 2, 51, 17, 2, 2), (128, 2, 16, 4, 2, 3),
                                                      (16384, 2, 24, 12, 1, 2)])
def test_whole_preimage_replays_on_the_cpu(gpu, oracle, monkeypatch, n, depth, bits, base, d, cols, assembly):
    """Trapdoor generation and a preimage call with the OS seeds replaced by fixed ones (the test-only
    hook mxx_amd.sampler.seed_source): R, E, A and the preimage x are bit-identical to the CPU chain
    oracle.trapdoor_gen / oracle.preimage (restating src/sampler/trapdoor/gpu.rs:202-369).  Both assemblies of the
    result: the small-operand one (products and sums as the reference forms them) and the large-operand one (stacked
    left factor over p2, NTT(z) + p2 in one pass, the top block from the output's own rows; 2^14 takes the fused kernel)."""
    import mxx_amd.trapdoor as trapdoor_mod
    from mxx_amd.sampler import seed_source

    monkeypatch.setattr(trapdoor_mod, "TRAFFIC_BOUND_BYTES", 0 if assembly == "large" else 1 << 62)

    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    sigma = 4.578
    master = bytes(range(32))
    seeds = [oracle._seed_from(master, i).tobytes() for i in range(6)]  # r, e, a_bar | p2, p1, z
    r, e, a = oracle.trapdoor_gen(moduli, n, base, sigma, d, master)
    target = oracle.matrix_ntt(rand_matrix(oracle, 77, d, cols, moduli, n), moduli)
    want = oracle.preimage(moduli, n, base, sigma, r, e, a, target, master)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, sigma)
    with seed_source(seeds):
        td, A = sampler.trapdoor(p, d)
        assert np.array_equal(td.r.to_rns(), r) and np.array_equal(td.e.to_rns(), e) and np.array_equal(A.to_rns(), a)
        gt = gpu.GpuDCRTPolyMatrix.from_rns(p, target, True)
        x = sampler.preimage(p, td, A, gt)
    assert np.array_equal(x.ensure_eval().to_rns(), want)
    assert A * x == gt


@pytest.mark.parametrize("d,n,bits", [(1, 64, 24), (2, 32, 24), (1, 32, 51), (5, 16, 24)])
@pytest.mark.parametrize("per_lane", ["", "3"])
def test_p1_sampler_bit_exact(gpu, oracle, hip_env, per_lane, d, n, bits):
    if per_lane:  # several elements per lane in the persistent-lane form (m <= 4)
        hip_env.set("MXX_HIP_SAMPLER_PER_LANE", per_lane)
    depth, base = 2, 12 if bits == 24 else 17
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    rng = np.random.default_rng(d * 100 + n)

    def small(rows, cols, bound):
        v = rng.integers(-bound, bound + 1, size=(rows, cols, n))
        return np.stack([np.mod(v, q).astype(np.uint64) for q in moduli], axis=-2)

    a, b, dm = small(d, d, 60), small(d, d, 25), small(d, d, 60)
    sigma, s_par, dgg = 12.0, 900.0, 4.578
    cols = 3
    tp2 = small(2 * d, cols, 5000)
    ga, gb, gd = (gpu.GpuDCRTPolyMatrix.from_rns(p, x, False) for x in (a, b, dm))
    gtp2 = gpu.GpuDCRTPolyMatrix.from_rns(p, tp2, False)
    cache = gpu.GpuDCRTPolyMatrix.create_p1_covariance_cache(ga, gb, gd, sigma, s_par, dgg)
    s = seed(gpu, 9)
    out = gpu.GpuDCRTPolyMatrix.sample_p1_full_cached(cache, gtp2, s)
    assert out.is_ntt and out.size() == (2 * d, cols)
    sv, up = oracle.p1_covariance(a, b, dm, moduli, sigma, s_par, dgg)
    c_scale = -(sigma * sigma) / (s_par * s_par - sigma * sigma)
    want = oracle.sample_p1(tp2, moduli, sv, up, c_scale, s)
    assert np.array_equal(out.to_coeff_rns(), want)


@pytest.mark.parametrize("n,depth,bits,base,d,cols", [
    (128, 2, 17, 1, 1, 1),     # square
    (128, 2, 16, 8, 2, 5),     # wide target, base 8... 2^8
    (128, 2, 16, 4, 3, 2),     # narrow
    (256, 2, 24, 12, 1, 4),    # bench-style base
    (64, 2, 51, 17, 2, 3),     # u64 words
])
def test_trapdoor_and_preimage(gpu, oracle, n, depth, bits, base, d, cols):
    """A*[R;E;I] == G and A*preimage == target (trapdoor/gpu.rs:547-663), norm bound (:690-811)."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    sigma = 4.578
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, sigma)
    td, A = sampler.trapdoor(p, d)
    k = p.modulus_digits()
    assert A.size() == (d, d * (k + 2))
    I = gpu.GpuDCRTPolyMatrix.identity(p, d * k)
    rei = td.r.concat_rows([td.e, I])
    assert A * rei == gpu.GpuDCRTPolyMatrix.gadget_matrix(p, d)
    us = gpu.GpuDCRTPolyUniformSampler()
    target = us.sample_uniform(p, d, cols, gpu.DistType.FinRingDist())
    x = sampler.preimage(p, td, A, target)
    assert x.size() == (d * (k + 2), cols)
    assert A * x == target
    # infinity-norm bound on CRT-reconstructed, centred coefficients
    Q = p.modulus()
    coeffs = x.coeffs()
    worst = max(min(v, Q - v) for row in coeffs for poly in row for v in poly)
    c = ((1 << base) + 1) * sigma
    s_par = 1.8 * ((1 << base) + 1) * sigma * sigma * (math.sqrt(d * n * k) + math.sqrt(2 * n) + 4.7)
    bound = 6.5 * s_par + 6.5 * math.sqrt(d * k * n) * 6.0 * sigma * c  # generous: |p| + |[R;E] z|
    assert worst < bound
    assert worst > 0
    # not the deterministic gadget solution: two calls differ
    x2 = sampler.preimage(p, td, A, target)
    assert not (x2 == x) and A * x2 == target


def test_preimage_extend(gpu, oracle):
    p = make_params(gpu, oracle, 128, 2, 16, 4)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, A = sampler.trapdoor(p, 2)
    us = gpu.GpuDCRTPolyUniformSampler()
    ext = us.sample_uniform(p, 2, 3, gpu.DistType.FinRingDist())
    target = us.sample_uniform(p, 2, 2, gpu.DistType.FinRingDist())
    x = sampler.preimage_extend(p, td, A, ext, target)
    assert A.concat_columns([ext]) * x == target


@pytest.mark.parametrize("n,depth,bits,base", [(128, 2, 17, 4), (64, 2, 51, 17), (16384, 3, 24, 12), (1024, 3, 24, 7)])
def test_sample_decomposed_extension_equals_sample_then_decompose(gpu, oracle, n, depth, bits, base):
    """gpupoly_matrix_sample_decomposed == sample_distribution(...).decompose() / .small_decompose() for every
    distribution (src/sampler/gpu.rs:91-115), EVAL and bit for bit."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    key = bytes(range(32))
    hs = gpu.GpuDCRTPolyHashSampler()
    for dist in (gpu.DistType.FinRingDist(), gpu.DistType.GaussDist(4.578), gpu.DistType.BitDist(), gpu.DistType.TernaryDist()):
        a = hs.sample_hash(p, key, b"dec", 2, 3, dist)
        full = hs.sample_hash_decomposed(p, key, b"dec", 2, 3, dist)
        small = hs.sample_hash_small_decomposed(p, key, b"dec", 2, 3, dist)
        assert full.is_ntt and small.is_ntt
        assert full == a.decompose()
        assert small == a.small_decompose()
    assert hs.sample_hash_decomposed(p, key, b"dec", 0, 3, gpu.DistType.BitDist()).size() == (0, 3)


@pytest.mark.parametrize("n,depth,bits,base", [(16, 3, 18, 6), (1024, 2, 24, 12), (64, 2, 51, 17)])
@pytest.mark.parametrize("dist,sigma", [("uniform", 0.0), ("gauss", 4.578), ("gauss", 8191.5), ("bit", 0.0), ("ternary", 0.0)])
def test_rng_compat_reference_keying(gpu, oracle, hip_env, n, depth, bits, base, dist, sigma):
    """MXX_HIP_RNG_COMPAT=reference: gpu_matrix_sample_distribution(_columns) keyed and consumed
    exactly as the reference's device RNG (cuda/src/ChaCha.cu:104-167, cuda/src/matrix/MatrixSampling.cu:30-147,239-289) -
    bit for bit against the CPU restatement of that keying; column windows still commute (src/sampler/gpu.rs:323-361);
    with the switch unset the default keying is back, and the two differ."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    s = seed(gpu, 21)
    code = oracle.DIST[dist]
    default = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, sigma, s)
    hip_env.set("MXX_HIP_RNG_COMPAT", "reference")
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, sigma, s)
    assert m.is_ntt
    want = oracle.sample_distribution_refkey(2, 3, moduli, n, dist, sigma, s)
    assert np.array_equal(m.to_coeff_rns(), want)
    w = gpu.GpuDCRTPolyMatrix.sample_distribution_columns(p, 2, 3, 1, 2, code, sigma, s)
    assert w == m.slice(0, 2, 1, 3)
    assert not (m == default)
    hip_env.unset("MXX_HIP_RNG_COMPAT")
    assert gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 3, code, sigma, s) == default
    assert np.array_equal(default.to_coeff_rns(), oracle.sample_distribution(2, 3, moduli, n, dist, sigma, s))


def test_rng_compat_hash_sampler_and_preimage_predicates(gpu, oracle, hip_env):
    """the callers of the seeded sampler under the reference keying: the hash sampler stays deterministic and window-
    consistent, trapdoor generation and a preimage keep their exact predicates (R, E, p2 come from the compat sampler)"""
    hip_env.set("MXX_HIP_RNG_COMPAT", "reference")
    p = make_params(gpu, oracle, 256, 2, 24, 12)
    hs = gpu.GpuDCRTPolyHashSampler()
    key = bytes(range(32))
    a = hs.sample_hash(p, key, b"tag", 2, 4, gpu.DistType.FinRingDist())
    assert a == hs.sample_hash(p, key, b"tag", 2, 4, gpu.DistType.FinRingDist())
    assert hs.sample_hash_columns(p, key, b"tag", 2, 4, 1, 2, gpu.DistType.FinRingDist()) == a.slice(0, 2, 1, 3)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, A = sampler.trapdoor(p, 1)
    k = p.modulus_digits()
    assert A * td.r.concat_rows([td.e, gpu.GpuDCRTPolyMatrix.identity(p, k)]) == gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 1)
    target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, 3, gpu.DistType.FinRingDist())
    assert A * sampler.preimage(p, td, A, target) == target
