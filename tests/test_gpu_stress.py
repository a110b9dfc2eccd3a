"""GPU: allocation churn + large buffers.  Guards the stream-ordered caching allocator
(the ROCm 7.2 hipMallocAsync pool corrupted large copies under exactly this pattern)."""
import numpy as np
import pytest

from conftest import make_params

pytestmark = pytest.mark.gpu


def test_alloc_churn_large_matrices(gpu, oracle):
    n = 16384
    p = make_params(gpu, oracle, n, 4, 24, 12)
    us = gpu.GpuDCRTPolyUniformSampler()
    for it in range(6):
        z = us.sample_uniform(p, 16, 20, gpu.DistType.FinRingDist())  # 320 polys x 4 limbs = 84 MB
        c = z.clone()
        assert c == z
        host0 = z.to_rns()
        zc = z.clone()
        zc.intt_all_in_place()
        zc.ntt_all_in_place()
        assert zc == z
        assert np.array_equal(zc.to_rns(), host0)
        a = z.clone()
        a.intt_all_in_place()
        b = z.clone()
        b.intt_all_in_place()
        assert a == b
        del a, b, c, zc  # return blocks to the cache; the next iteration reuses them
    # full-size property of BASELINE config 1: INTT(NTT(x)) == x and linearity on 1024 x 4 vectors
    x = us.sample_uniform(p, 1024, 1, gpu.DistType.FinRingDist())
    y = us.sample_uniform(p, 1024, 1, gpu.DistType.FinRingDist())
    xs, ys = x.clone(), y.clone()
    xs.intt_all_in_place()
    ys.intt_all_in_place()
    s = xs + ys            # coefficient domain
    s.ntt_all_in_place()
    assert s == x + y      # NTT(a+b) == NTT(a)+NTT(b)


def test_preimage_bench_shape_repeated(gpu, oracle):
    """benches/bench_preimage_gpu.rs shape; every call must satisfy A*x == u exactly."""
    n = 16384
    p = make_params(gpu, oracle, n, 10, 24, 12)
    s = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, A = s.trapdoor(p, 1)
    us = gpu.GpuDCRTPolyUniformSampler()
    G = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 1)
    for it in range(4):
        target = us.sample_uniform(p, 1, 50, gpu.DistType.FinRingDist())
        z = target.clone().gauss_samp_gq_arb_base(s.c, s.sigma, gpu.random_gpu_rng_seed())
        assert G * z == target
        x = s.preimage(p, td, A, target)
        assert A * x == target
