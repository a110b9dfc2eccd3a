"""GPU: the library's launch trace (gpupoly_trace_begin / _end), region markers and the peer-access query - the
instruments bench.py composes its multi-kernel rooflines from."""
import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu


def test_launch_trace_names_bytes_and_durations(gpu, oracle):
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, 16384, 2, 24, 12)
    moduli, n = p.moduli(), 16384
    a = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 5, 2, 3, moduli, n), False)
    b = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 6, 3, 4, moduli, n), False)
    gpu.gpu_device_sync()
    _ffi.trace_begin()
    a.ntt_all_in_place()
    b.ntt_all_in_place()
    c = a * b
    d = c + c
    e = d.clone()
    gpu.gpu_device_sync()
    tr = _ffi.trace_end()
    names = [t["kernel"] for t in tr]
    poly = 2 * n * 4  # bytes of one polynomial (2 limbs, u32)
    assert len(tr) == 5, names
    assert "ntt14::fwd_kernel" in names[0] and tr[0]["bytes"] == 2 * 6 * poly
    assert tr[1]["bytes"] == 2 * 12 * poly
    assert "matmul" in names[2] and tr[2]["bytes"] == (6 + 12 + 8) * poly
    assert "elementwise_kernel" in names[3] and tr[3]["bytes"] == 2 * 8 * poly  # c + c: one operand read, one written
    assert "copy" in names[4] and tr[4]["bytes"] == 2 * 8 * poly
    assert all(0.0 < t["ms"] < 50.0 for t in tr)
    assert tr[2]["blocks"] > 0 and tr[2]["threads"] in (64, 128, 256)
    # tracing is off again: nothing accumulates, results unaffected
    f = a * b
    assert _ffi.trace_end() == []
    assert f == c and e == d
    want = oracle.matmul(oracle.matrix_ntt(rand_matrix(oracle, 5, 2, 3, moduli, n), moduli),
                         oracle.matrix_ntt(rand_matrix(oracle, 6, 3, 4, moduli, n), moduli), moduli)
    assert np.array_equal(c.to_rns(), want)


def test_trace_covers_a_preimage_call(gpu, oracle):
    """every kernel of a preimage call appears with a duration; the samplers and the fused transform state their bytes"""
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, 1024, 3, 24, 12)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, pub = sampler.trapdoor(p, 1)
    target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, 3, gpu.DistType.FinRingDist())
    sampler.preimage(p, td, pub, target)
    n0 = _ffi.lib().gpupoly_launch_count()
    _ffi.trace_begin()
    x = sampler.preimage(p, td, pub, target)
    gpu.gpu_device_sync()
    tr = _ffi.trace_end()
    launches = _ffi.lib().gpupoly_launch_count() - n0
    kernels = [t for t in tr if "copy" not in t["kernel"]]
    assert len(kernels) == launches
    names = " ".join(t["kernel"] for t in tr)
    for must in ("sample_gauss_kernel", "gauss_samp_lanes_kernel", "p1_sample_lanes_kernel", "scatter_i64_kernel", "matmul"):
        assert must in names, (must, names)
    stated = {t["kernel"].split("<")[0]: t["bytes"] for t in tr if t["bytes"]}
    assert stated["sample_gauss_kernel"] == 6 * 3 * 1024 * 8  # p2: d k x cols polynomials of int64 samples
    assert pub * x == target


def test_markers_and_peer_access(gpu, oracle):
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, 1024, 3, 24, 12)
    n0 = _ffi.lib().gpupoly_launch_count()
    p.ctx().marker(1)
    p.ctx().marker(2)
    gpu.gpu_device_sync()
    assert _ffi.lib().gpupoly_launch_count() == n0  # markers are not part of the library's launch count
    m = _ffi.peer_access_matrix()
    k = gpu.detected_gpu_device_count()
    assert len(m) == k and all(len(r) == k and r[i] == 1 for i, r in enumerate(m))
