"""What one rank sees under bench.py's column-block partition of M2A (n = 2^14, L = 15, (1 x 30)(30 x 120/N)):
time per product against the 5.8 TB/s the full shape streams at."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L = 16384, 15
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
us = mx.GpuDCRTPolyUniformSampler()
d = mx.DistType.FinRingDist()
a = us.sample_uniform(p, 1, 30, d)
for world in (1, 2, 4, 8):
    c = 120 // world
    b = us.sample_uniform(p, 30, c, d)
    res = {}
    for path in ("reg", "lds", ""):
        if path:
            os.environ["MXX_HIP_MATMUL_PATH"] = path
        else:
            os.environ.pop("MXX_HIP_MATMUL_PATH", None)
        _ffi.reload_env()
        out = a * b
        mx.gpu_device_sync()
        ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(10):
                out = a * b
            ts.append(ctx.timer_stop() / 10)
        res[path or "auto"] = min(ts)
    gb = (30 * c + 30 + c) * L * n * 4 / 1e9
    print(f"N={world}: (1x30)(30x{c})  auto {res['auto']*1e3:7.1f} us  reg {res['reg']*1e3:7.1f}  lds {res['lds']*1e3:7.1f}   {gb/res['auto']:.2f} TB/s auto; ideal at 5.8 TB/s {gb/5.8*1e3:.1f} us", flush=True)
    del b
