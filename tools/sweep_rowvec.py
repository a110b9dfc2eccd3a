"""Row-vector products (rows = 1..3): register tile (with the loads-ahead form on small grids) against the 64-slot
LDS tile and the automatic choice.  Several ring sizes / limb counts, since the rule depends on the grid size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

us = mx.GpuDCRTPolyUniformSampler()
d = mx.DistType.FinRingDist()
bad = 0
for n, L in ((16384, 15), (16384, 4), (4096, 8), (1024, 4)):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
    ctx = p.ctx()
    for r in (1, 2, 3):
        for k in (16, 30, 64, 256):
            for c in (4, 8, 15, 16, 30, 60, 120):
                if (r * k + k * c + r * c) * L * n * 4 > 6e9:
                    continue
                a, b = us.sample_uniform(p, r, k, d), us.sample_uniform(p, k, c, d)
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
                    for _ in range(3):
                        ctx.timer_start()
                        for _ in range(5):
                            out = a * b
                        ts.append(ctx.timer_stop() / 5)
                    res[path or "auto"] = min(ts)
                best = "reg" if res["reg"] <= res["lds"] else "lds"
                ratio = res["auto"] / res[best]
                flag = "  <-- auto %.0f %% slower than %s" % ((ratio - 1) * 100, best) if ratio > 1.08 else ""
                bad += bool(flag)
                print(f"n={n} L={L} ({r}x{k})*({k}x{c}): auto {res['auto']*1e3:8.1f} us  reg {res['reg']*1e3:8.1f}  lds {res['lds']*1e3:8.1f}{flag}", flush=True)
                del a, b, out
print("shapes where auto loses more than 8 %:", bad)
