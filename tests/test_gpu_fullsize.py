"""GPU: size-independent properties at BASELINE.json's full shapes (the oracle cannot finish these
sizes in seconds, so parity is carried by exact algebraic identities on device-sampled inputs).

  M2a  bench_matrix_mul shape  (n=2^14, L=15, 1x30 * 30x120): distributivity and scalar associativity
  M2b  config 3                (n=2^14, L=8, 64x64): G * G^-1(M) == M through the fused decompose and the
                               streamed product with inner dimension 1024; (A*B)*C == A*(B*C)
  M3a  bench_preimage shape    (n=2^14, L=10, 50 columns): A*x == u and the limb-0 infinity norm of x
  M3b  config 3 of BASELINE    (the same at L=8)
"""
import math

import numpy as np
import pytest

from conftest import make_params

pytestmark = pytest.mark.gpu

N = 16384


def test_m2a_distributive_and_scalar_associative(gpu, oracle):
    p = make_params(gpu, oracle, N, 15, 24, 12)
    us = gpu.GpuDCRTPolyUniformSampler()
    u = gpu.DistType.FinRingDist()
    a = us.sample_uniform(p, 1, 30, u)
    b1 = us.sample_uniform(p, 30, 120, u)
    b2 = us.sample_uniform(p, 30, 120, u)
    assert a * (b1 + b2) == a * b1 + a * b2
    s = us.sample_poly(p, u)
    assert (a * b1) * s == a * (b1 * s)
    assert not (a * b1 == a * b2)


def test_m2b_gadget_inverse_and_associativity(gpu, oracle):
    p = make_params(gpu, oracle, N, 8, 24, 12)
    us = gpu.GpuDCRTPolyUniformSampler()
    u = gpu.DistType.FinRingDist()
    m = us.sample_uniform(p, 64, 64, u)
    dec = m.decompose()                       # 1024 x 64, fused digit + NTT kernel
    k = p.modulus_digits()
    assert dec.size() == (64 * k, 64) and dec.is_ntt
    g = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 64)
    assert g * dec == m                       # inner dimension 1024
    del dec, g
    a = us.sample_uniform(p, 64, 64, u)
    b = us.sample_uniform(p, 64, 64, u)
    assert (a * b) * m == a * (b * m)
    # digits are small: every coefficient of G^-1(M) is below the base in every limb
    d_small = us.sample_uniform(p, 2, 2, u).decompose().to_coeff_rns()
    assert int(d_small.max()) < (1 << 12)


def test_m2a_output_columns_against_the_oracle(gpu, oracle):
    """The bench shape's product, three of its 120 output columns recomputed by the CPU restatement (30 ring products
    each, all 15 limbs): an error shared by both sides of an algebraic identity - a wrong per-limb constant at L = 15,
    say - cannot hide here."""
    p = make_params(gpu, oracle, N, 15, 24, 12)
    moduli = p.moduli()
    us = gpu.GpuDCRTPolyUniformSampler()
    a = us.sample_uniform(p, 1, 30, gpu.DistType.FinRingDist())
    b = us.sample_uniform(p, 30, 120, gpu.DistType.FinRingDist())
    c = a * b
    assert "matmul_kernel<u32,1,8,4" in p.ctx().last_kernel()  # the kernel bench.py times
    a_h = a.to_rns()
    for col in (0, 57, 119):
        want = oracle.matmul(a_h, b.slice_columns(col, col + 1).to_rns(), moduli)
        assert np.array_equal(c.slice_columns(col, col + 1).to_rns(), want), col


def test_m2b_entries_and_decompose_rows_against_the_oracle(gpu, oracle):
    """64^3 at L = 8 through the streamed product: three output entries (64 ring products each) against the CPU
    restatement; and one source entry of the 64 x 64 -> 1024 x 64 decomposition (its 16 digit polynomials, EVAL form)
    against oracle.decompose + oracle NTT."""
    p = make_params(gpu, oracle, N, 8, 24, 12)
    moduli = p.moduli()
    us = gpu.GpuDCRTPolyUniformSampler()
    a = us.sample_uniform(p, 64, 64, gpu.DistType.FinRingDist())
    b = us.sample_uniform(p, 64, 64, gpu.DistType.FinRingDist())
    c = a * b
    assert "mmdma32" in p.ctx().last_kernel()
    for r, col in ((0, 0), (37, 21), (63, 63)):
        want = oracle.matmul(a.slice_rows(r, r + 1).to_rns(), b.slice_columns(col, col + 1).to_rns(), moduli)
        assert np.array_equal(c.slice(r, r + 1, col, col + 1).to_rns(), want), (r, col)
    k = p.modulus_digits()
    dec = a.decompose()
    for r, col in ((0, 0), (41, 63)):
        src = oracle.matrix_ntt(a.slice(r, r + 1, col, col + 1).to_rns(), moduli, inverse=True)
        want = oracle.matrix_ntt(oracle.decompose(src, moduli, 12), moduli)
        assert want.shape[0] == k
        assert np.array_equal(dec.slice(r * k, (r + 1) * k, col, col + 1).to_rns(), want), (r, col)


@pytest.mark.parametrize("depth", [10, 8])  # bench_preimage_gpu.rs shape (M3a) and BASELINE configs[3] (M3b, L = 8)
def test_m3_preimage_relation_and_norm(gpu, oracle, depth):
    p = make_params(gpu, oracle, N, depth, 24, 12)
    sigma, base = 4.578, 12
    from mxx_amd.sampler import seed_source

    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, sigma)
    k = p.modulus_digits()
    # fixed seeds (R, E, A_bar | target | p2, p1, z): the bound below is a 6.5-sigma event over 18 million entries - about one
    # run in a thousand with OS seeds (it happened once in round 5) - so the draw is pinned; the statistics of the samplers
    # are tested at scale in test_gpu_sampler_stats.py
    with seed_source(bytes((13 * j + 3 * i + depth) & 0xFF for i in range(32)) for j in range(7)):
        td, A = sampler.trapdoor(p, 1)
        target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, 50, gpu.DistType.FinRingDist())
        x = sampler.preimage(p, td, A, target)
    assert x.size() == (k + 2, 50)
    assert A * x == target
    # x is one integer vector, |x| < 6.5 s ~ 2^30 > q_i/2: rebuild it from limbs 0 and 1 (Garner), centre it
    # modulo q0*q1, and check that limb 2 holds the same integer
    xr = x.to_coeff_rns()
    q0, q1, q2 = (int(q) for q in p.moduli()[:3])
    r0, r1, r2 = (xr[:, :, l].astype(np.int64) for l in range(3))
    t = ((r1 - r0) % q1 * pow(q0, -1, q1)) % q1
    v = r0 + q0 * t
    v = np.where(v > (q0 * q1) // 2, v - q0 * q1, v)
    assert np.array_equal(v % q2, r2)
    c = ((1 << base) + 1) * sigma
    s_par = 1.8 * ((1 << base) + 1) * sigma * sigma * (math.sqrt(N * k) + math.sqrt(2 * N) + 4.7)
    # rows 0..1 carry p1 + [R;E] z, the rest p2 + z with p2 of width sqrt(s^2 - c^2)
    assert np.abs(v).max() < 6.5 * s_par
    width = math.sqrt(s_par * s_par - c * c)
    assert 0.9 * width < v[2:].std() < 1.1 * width


@pytest.mark.parametrize("logn,bits,polys", [(13, 24, 8192), (14, 24, 4096), (15, 28, 2048), (12, 51, 8192)])
def test_ntt_batches_beyond_the_infinity_cache(gpu, oracle, logn, bits, polys):
    """Batches of at least 1 GiB take the non-temporal forms of the whole-vector LDS kernels and of the grouped 2^14
    kernels (ntt_lds_dispatch.inc):
    round trip on the device, the first and last polynomials against the CPU restatement, both directions."""
    n = 1 << logn
    moduli = oracle.gen_crt_basis(n, 4, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 12)
    assert polys * 4 * n * p.ctx().word_bytes() >= 1 << 30
    m = gpu.GpuDCRTPolyUniformSampler().sample_uniform(p, polys, 1, gpu.DistType.FinRingDist())  # EVAL form
    ev = m.clone()
    m.intt_all_in_place()
    for r in (0, polys - 1):
        got = m.slice_rows(r, r + 1).to_rns()
        assert np.array_equal(got, oracle.matrix_ntt(ev.slice_rows(r, r + 1).to_rns(), moduli, inverse=True))
    m.ntt_all_in_place()
    assert m == ev


def test_u64_row_vector_product_beyond_the_infinity_cache(gpu, oracle):
    """64-bit words, a B operand larger than the Infinity Cache: the register tile streams it with non-temporal loads
    (arith.hip).  Distributivity, and every column against the one-column product (a B small enough to take the
    cacheable form of the kernel)."""
    n = 16384
    moduli = oracle.gen_crt_basis(n, 4, 51)
    p = gpu.GpuDCRTPolyParams(n, moduli, 17)
    us = gpu.GpuDCRTPolyUniformSampler()
    a1 = us.sample_uniform(p, 1, 16, gpu.DistType.FinRingDist())
    a2 = us.sample_uniform(p, 1, 16, gpu.DistType.FinRingDist())
    b = us.sample_uniform(p, 16, 40, gpu.DistType.FinRingDist())
    assert 16 * 40 * 4 * n * 8 > 1 << 28
    c1 = a1 * b
    assert (a1 + a2) * b == c1 + a2 * b
    for col in (0, 17, 39):
        assert a1 * b.slice_columns(col, col + 1) == c1.slice_columns(col, col + 1)


@pytest.mark.parametrize("depth", [10, 8])  # M3A (benches/bench_preimage_gpu.rs:7-56) and M3B (BASELINE configs[3])
def test_m3_whole_preimage_replays_on_the_cpu_at_full_size(gpu, oracle, depth):
    """the bit-exact replay AT the bench shape - n = 2^14, L = 10 / 8, d = 1, all 50 target columns,
    the large-operand assembly the bench times (stacked left factor over p2, NTT(z) + p2 through ntt14::fwd_add_kernel,
    the top block from the output's own rows).  R, E, A and every residue of the 22 x 50 / 18 x 50 preimage equal
    oracle.trapdoor_gen / oracle.preimage with the OS seeds replaced by fixed ones (test-only hook seed_source)."""
    from mxx_amd.sampler import seed_source

    p = make_params(gpu, oracle, N, depth, 24, 12)
    moduli = p.moduli()
    sigma, base, d, cols = 4.578, 12, 1, 50
    master = bytes((11 * i + depth) & 0xFF for i in range(32))
    seeds = [oracle._seed_from(master, i).tobytes() for i in range(6)]  # r, e, a_bar | p2, p1, z
    r, e, a = oracle.trapdoor_gen(moduli, N, base, sigma, d, master)
    target = oracle.matrix_ntt(oracle.random_matrix(770 + depth, d, cols, moduli, N), moduli)
    want = oracle.preimage(moduli, N, base, sigma, r, e, a, target, master)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, sigma)
    with seed_source(seeds):
        td, A = sampler.trapdoor(p, d)
        assert np.array_equal(td.r.to_rns(), r) and np.array_equal(td.e.to_rns(), e) and np.array_equal(A.to_rns(), a)
        gt = gpu.GpuDCRTPolyMatrix.from_rns(p, target, True)
        n0 = p.ctx().last_kernel()
        x = sampler.preimage(p, td, A, gt)
    del n0
    k = p.modulus_digits()
    assert x.size() == (k + 2, cols) and x.is_ntt
    got = x.to_rns()
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert A * x == gt


def test_m1_step_full_batch_against_the_oracle(gpu, oracle):
    """BASELINE configs[1] at full size through the kernels bench.py times: x <- INTT(NTT(x) o w) on 1024 polys x 4
    limbs - ntt14::fwd_kernel, then the fused product + inverse transform (gpupoly_matrix_mul_scalar_intt) - and 40
    polynomials spread over the batch (first, last, every 27th) against oracle.matrix_ntt / oracle.pointwise."""
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, N, 4, 24, 12)
    moduli = p.moduli()
    us = gpu.GpuDCRTPolyUniformSampler()
    x = us.sample_uniform(p, 1024, 1, gpu.DistType.FinRingDist())
    x.intt_all_in_place()
    w = us.sample_uniform(p, 1, 1, gpu.DistType.FinRingDist())
    rows = sorted(set([0, 1023] + list(range(5, 1024, 27))))
    assert len(rows) >= 32
    x_h = {r: x.slice_rows(r, r + 1).to_rns() for r in rows}  # COEFF
    w_h = w.to_rns()
    lib = _ffi.lib()
    _ffi.check_status(lib.gpu_matrix_ntt_all(x.raw), "gpu_matrix_ntt_all")
    x.is_ntt = True
    ev = {r: x.slice_rows(r, r + 1).to_rns() for r in (0, 518, 1023)}
    _ffi.check_status(lib.gpupoly_matrix_mul_scalar_intt(x.raw, x.raw, w.raw), "gpupoly_matrix_mul_scalar_intt")
    x.is_ntt = False
    for r in rows:
        e = oracle.matrix_ntt(x_h[r], moduli)
        if r in ev:
            assert np.array_equal(ev[r], e), r
        want = oracle.matrix_ntt(oracle.pointwise("mul", e, w_h, moduli), moduli, inverse=True)
        assert np.array_equal(x.slice_rows(r, r + 1).to_rns(), want), r


def test_m4_chain_step_replays_on_the_cpu_at_depth_12(gpu, oracle):
    """BASELINE configs[4] parameters (n = 256, 12 limbs of 51 bits, base 2^17, d = 2) - the step bench.py's chain_m4 block
    times: preimage of a 4-column target (bit for bit against oracle.preimage, fixed seeds), the encoding times the key,
    and mul_decompose, each against the CPU restatement."""
    from mxx_amd.sampler import seed_source

    n, depth, bits, base, d = 256, 12, 51, 17, 2
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    k = p.modulus_digits()
    master = bytes((5 * i + 1) & 0xFF for i in range(32))
    seeds = [oracle._seed_from(master, i).tobytes() for i in range(6)]
    r, e, a = oracle.trapdoor_gen(moduli, n, base, 4.578, d, master)
    target = oracle.matrix_ntt(oracle.random_matrix(4004, d, 2 * d, moduli, n), moduli)
    want = oracle.preimage(moduli, n, base, 4.578, r, e, a, target, master)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    with seed_source(seeds):
        td, A = sampler.trapdoor(p, d)
        K = sampler.preimage(p, td, A, gpu.GpuDCRTPolyMatrix.from_rns(p, target, True))
    assert np.array_equal(A.to_rns(), a)
    assert np.array_equal(K.ensure_eval().to_rns(), want)
    c0 = oracle.matrix_ntt(oracle.random_matrix(4005, 1, A.col_size(), moduli, n), moduli)
    g0 = gpu.GpuDCRTPolyMatrix.from_rns(p, c0, True)
    assert np.array_equal((g0 * K).to_rns(), oracle.matmul(c0, want, moduli))
    B = oracle.matrix_ntt(oracle.random_matrix(4006, d, d * k, moduli, n), moduli)
    M = oracle.random_matrix(4007, d, 3, moduli, n)  # COEFF
    got = gpu.GpuDCRTPolyMatrix.from_rns(p, B, True).mul_decompose(gpu.GpuDCRTPolyMatrix.from_rns(p, oracle.matrix_ntt(M, moduli), True))
    dec = oracle.matrix_ntt(oracle.decompose(M, moduli, base), moduli)
    assert np.array_equal(got.to_rns(), oracle.matmul(B, dec, moduli))
