"""Is the automatic kernel choice of gpu_matrix_mul near the best forced family?  n = 2^14, L = 4, many shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L = 16384, 4
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
us = mx.GpuDCRTPolyUniformSampler()
d = mx.DistType.FinRingDist()

def timed(a, b):
    out = a * b
    mx.gpu_device_sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start(); out = a * b; ms = ctx.timer_stop(); best = min(best, ms)
    return best

bad = 0
for r in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
    for k in (16, 64, 256):
        for c in (4, 8, 16, 32, 64, 120):
            if r * k + k * c + r * c > 40000:
                continue
            a, b = us.sample_uniform(p, r, k, d), us.sample_uniform(p, k, c, d)
            res = {}
            for path in ("reg", "lds", "dma", "wide", ""):  # auto last: the first product after sampling runs 5-9 % slow
                if path:
                    os.environ["MXX_HIP_MATMUL_PATH"] = path
                else:
                    os.environ.pop("MXX_HIP_MATMUL_PATH", None)
                _ffi.reload_env()
                res[path or "auto"] = timed(a, b)
            best = min(res, key=lambda x: res[x] if x != "auto" else 1e9)
            ratio = res["auto"] / res[best]
            flag = "  <-- auto is %.0f %% slower than %s" % ((ratio - 1) * 100, best) if ratio > 1.08 else ""
            bad += bool(flag)
            print(f"({r}x{k})*({k}x{c}): auto {res['auto']:7.3f}  reg {res['reg']:7.3f}  lds {res['lds']:7.3f}  dma {res['dma']:7.3f}  wide {res['wide']:7.3f}{flag}", flush=True)
            del a, b
print("shapes where auto loses more than 8 %:", bad)
