"""GPU: BASELINE.json config 5 flavour — the parameter family of the reference's
tests/test_gpu_ggh15_modp_chain.rs (n=256, 51-bit limbs, base 2^17, depth <= 12) exercised as
the chain of hot-path calls those schemes make: trapdoor -> preimage -> mul_decompose -> add,
on the uint64 word path, every step held to its exact predicate."""
import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [3, 6, 12])  # 12: tests/test_gpu_ggh15_modp_chain.rs:36-42, what `bench.py --workload m4` times
def test_preimage_mul_decompose_chain_u64(gpu, oracle, depth):
    n, bits, base, d = 256, 51, 17, 2
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    assert p.ctx().word_bytes() == 8
    k = p.modulus_digits()
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    us = gpu.GpuDCRTPolyUniformSampler()
    G = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, d)
    td0, A0 = sampler.trapdoor(p, d)
    td1, A1 = sampler.trapdoor(p, d)
    # level 0 -> 1: K with A0 * K == A1-slice (a GGH15-style transition key), 2 target blocks
    target = A1.slice(0, d, 0, 2 * d)
    K = sampler.preimage(p, td0, A0, target)
    assert A0 * K == target
    # an encoding s*A0 + e pushed through K, compared with the oracle on the host
    s = us.sample_uniform(p, 1, d, gpu.DistType.BitDist())
    e = us.sample_uniform(p, 1, A0.col_size(), gpu.DistType.GaussDist(3.2))
    c0 = s * A0 + e
    c1 = c0 * K
    want = oracle.matmul(c0.to_rns(), K.to_rns(), moduli)
    assert np.array_equal(c1.to_rns(), want)
    # s*A0*K == s*target, so c1 - s*target == e*K (small): exact identity in R_q
    assert c1 - s * target == e * K
    # mul_decompose: B (d*k x cols) times G^-1(M) with a public matrix M
    B = us.sample_uniform(p, d, d * k, gpu.DistType.FinRingDist())
    M = us.sample_uniform(p, d, 3, gpu.DistType.FinRingDist())
    got = B.mul_decompose(M)
    Mc = M.to_coeff_rns()
    dec = oracle.matrix_ntt(oracle.decompose(Mc, moduli, base), moduli)
    assert np.array_equal(got.to_rns(), oracle.matmul(B.to_rns(), dec, moduli))
    # gadget homomorphism used by BGG+: (G * G^-1(M)) == M
    assert G * M.decompose() == M
    # compact wire format survives the u64 path
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, K.to_compact_bytes()) == K
