"""Fat products at n = 2^14, L = 8 under each streamed kernel (MXX_HIP_MATMUL_PATH = dma | wide) and the automatic choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L = 16384, 8
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
us = mx.GpuDCRTPolyUniformSampler()
shapes = [tuple(int(x) for x in s.split("x")) for s in sys.argv[1:]] or [(64, 64, 64), (64, 256, 64), (128, 64, 128), (64, 64, 32), (128, 64, 64), (256, 16, 256), (56, 64, 64), (64, 64, 96)]
for (r, k, c) in shapes:
    a = us.sample_uniform(p, r, k, mx.DistType.FinRingDist())
    b = us.sample_uniform(p, k, c, mx.DistType.FinRingDist())
    res = {}
    for path in ("dma", "wide", ""):
        if path:
            os.environ["MXX_HIP_MATMUL_PATH"] = path
        else:
            os.environ.pop("MXX_HIP_MATMUL_PATH", None)
        _ffi.reload_env()
        out = a * b
        mx.gpu_device_sync()
        best = 1e9
        for _ in range(4):
            ctx.timer_start(); out = a * b; ms = ctx.timer_stop(); best = min(best, ms)
        res[path or "auto"] = best
    gb = (r * k + k * c + r * c) * L * n * 4 / 1e9
    print(f"({r}x{k})*({k}x{c}): " + "  ".join(f"{t} {v:7.3f}" for t, v in res.items()) + f" ms;  {gb:5.1f} GB algorithmic -> {gb / res['auto']:5.2f} TB/s auto", flush=True)
    del a, b, out
