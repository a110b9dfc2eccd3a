"""Transforms with 28-bit limbs (the reference's end-to-end default, n = 2^16) against 24-bit limbs and against the
fully reduced kernels the 28-bit case used to fall back to: ms per transform of `polys` x 4 limbs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

us = mx.GpuDCRTPolyUniformSampler()
for logn, polys in ((12, 2048), (14, 1024), (16, 256), (17, 128)):
    n = 1 << logn
    for bits in (24, 28):
        p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 4, bits), 12)
        ctx = p.ctx()
        m = us.sample_uniform(p, polys, 1, mx.DistType.FinRingDist())
        gb = 2.0 * n * 4 * polys * 4 / 1e9
        res = {}
        for path in ("", "generic" if logn <= 15 else "global"):
            if path:
                os.environ["MXX_HIP_NTT_PATH"] = path
            else:
                os.environ.pop("MXX_HIP_NTT_PATH", None)
            _ffi.reload_env()
            for direction in ("inv", "fwd"):
                ts = []
                for _ in range(4):
                    if direction == "inv":
                        m.ntt_all_in_place(); mx.gpu_device_sync()
                        ctx.timer_start(); m.intt_all_in_place(); ts.append(ctx.timer_stop())
                    else:
                        m.intt_all_in_place(); mx.gpu_device_sync()
                        ctx.timer_start(); m.ntt_all_in_place(); ts.append(ctx.timer_stop())
                res[(path or "tuned", direction)] = min(ts)
        os.environ.pop("MXX_HIP_NTT_PATH", None)
        _ffi.reload_env()
        print(f"n=2^{logn} {bits}-bit, {polys} polys x 4 limbs ({gb:.2f} GB): " + "  ".join(f"{k[0]} {k[1]} {v*1e3:7.1f} us ({gb/v:5.2f} TB/s)" for k, v in res.items()), flush=True)
        del m
