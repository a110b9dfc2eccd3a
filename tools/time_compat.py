"""Cost of MXX_HIP_RNG_COMPAT=reference (the reference device RNG's own keying, one thread per coefficient) against the
default samplers: uniform on M2A's 30 x 120 operand (L = 15) and the Gaussian p2 matrix of M3A (20 x 50, L = 10)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n = 16384
seed = mx.GpuRngSeed.from_bytes(bytes(range(32)))
for depth, rows, cols, dist, sigma, label in ((15, 30, 120, 0, 0.0, "uniform 30x120 L=15"), (10, 20, 50, 1, 1.17e8, "Gaussian 20x50 L=10")):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, depth, 24), 12)
    ctx = p.ctx()
    for compat in ("", "reference"):
        if compat:
            os.environ["MXX_HIP_RNG_COMPAT"] = compat
        else:
            os.environ.pop("MXX_HIP_RNG_COMPAT", None)
        _ffi.reload_env()
        ts = []
        for rep in range(4):
            ctx.timer_start()
            m = mx.GpuDCRTPolyMatrix.sample_distribution(p, rows, cols, dist, sigma, seed)
            ts.append(ctx.timer_stop())
            del m
        print(f"{label}, keying {compat or 'default'}: {statistics.median(ts[1:]):.2f} ms (sampler + its forward transform)")
