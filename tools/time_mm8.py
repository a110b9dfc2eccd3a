"""(8 x 1024) * (1024 x 64) at n = 2^14, L = 8 (the product inside mul_decompose of a 64 x 64) per kernel family."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L = 16384, 8
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
us = mx.GpuDCRTPolyUniformSampler()
for (r, k, c) in ((8, 1024, 64), (4, 1024, 64), (8, 256, 128), (16, 512, 64)):
    a = us.sample_uniform(p, r, k, mx.DistType.FinRingDist())
    b = us.sample_uniform(p, k, c, mx.DistType.FinRingDist())
    for path in ("", "reg", "lds", "dma"):
        if path:
            os.environ["MXX_HIP_MATMUL_PATH"] = path
        else:
            os.environ.pop("MXX_HIP_MATMUL_PATH", None)
        _ffi.reload_env()
        out = a * b
        mx.gpu_device_sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_start(); out = a * b; ms = ctx.timer_stop(); best = min(best, ms)
        gb = (r * k + k * c + r * c) * L * n * 4 / 1e9
        print(f"({r}x{k})*({k}x{c}) path={path or 'auto':4s}: {best:7.3f} ms  {gb / best:5.2f} TB/s algorithmic", flush=True)
    del a, b, out
