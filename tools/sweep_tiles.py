"""Register-tile shapes for 2..4-row products (MXX_HIP_MATMUL_TILE=RCS[p]): time per product, n = 2^14."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

us = mx.GpuDCRTPolyUniformSampler()
d = mx.DistType.FinRingDist()
n = 16384
for L, shapes in ((10, ((2, 20, 50), (3, 20, 50), (4, 20, 50), (2, 20, 7), (3, 20, 7))), (4, ((2, 64, 120), (3, 64, 120), (4, 64, 120), (3, 256, 60), (4, 256, 60), (3, 64, 16), (4, 64, 16))), (15, ((2, 30, 120), (3, 30, 120), (4, 30, 120)))):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
    ctx = p.ctx()
    for (r, k, c) in shapes:
        a, b = us.sample_uniform(p, r, k, d), us.sample_uniform(p, k, c, d)
        res = {}
        tiles = ["auto", "284", "284p", "244", "244p", "344", "344p", "444", "444p", "382", "382p", "481", "481p", "482", "482p"]
        for t in tiles:
            if t[0] != "a" and int(t[0]) < r and not (t[0] == "2" and r == 3):
                continue
            if t == "auto":
                os.environ.pop("MXX_HIP_MATMUL_TILE", None)
            else:
                os.environ["MXX_HIP_MATMUL_TILE"] = t
            _ffi.reload_env()
            out = a * b
            mx.gpu_device_sync()
            ts = []
            for _ in range(3):
                ctx.timer_start()
                for _ in range(5):
                    out = a * b
                ts.append(ctx.timer_stop() / 5)
            res[t] = min(ts) * 1e3
        gb = (r * k + k * c + r * c) * L * n * 4 / 1e9
        best = min((t for t in res if t != "auto"), key=lambda t: res[t])
        print(f"L={L} ({r}x{k})*({k}x{c}) {gb / res['auto'] * 1e3:5.2f} TB/s auto; best {best} {res[best]:.1f}: " + "  ".join(f"{t} {v:.1f}" for t, v in res.items()), flush=True)
        del a, b, out
os.environ.pop("MXX_HIP_MATMUL_TILE", None)
