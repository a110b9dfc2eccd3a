"""decompose with the digits fused into the forward transform's load against the two-step path (MXX_HIP_DECOMPOSE_FUSED=0)
across ring sizes and word widths; ~1 GB of digits each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

us = mx.GpuDCRTPolyUniformSampler()
for logn, L, bits, base in ((10, 5, 51, 17), (12, 4, 51, 17), (14, 4, 51, 17), (16, 4, 51, 17), (10, 8, 24, 12), (12, 8, 24, 12), (13, 8, 28, 14), (15, 8, 24, 12)):
    n = 1 << logn
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, bits), base)
    ctx = p.ctx()
    k = p.modulus_digits()
    polys = max(1, int(1e9 // (k * L * n * ctx.word_bytes())))
    rows = max(1, polys // 16)
    b = us.sample_uniform(p, rows, 16, mx.DistType.FinRingDist())
    res = {}
    for fused in ("1", "0"):
        os.environ["MXX_HIP_DECOMPOSE_FUSED"] = fused
        _ffi.reload_env()
        out = b.decompose(); mx.gpu_device_sync()
        ts = []
        for _ in range(4):
            ctx.timer_start(); out = b.decompose(); ts.append(ctx.timer_stop())
        res[fused] = min(ts)
        gb = out.row_size() * out.col_size() * L * n * ctx.word_bytes() / 1e9
        del out
    os.environ.pop("MXX_HIP_DECOMPOSE_FUSED", None)
    print(f"n=2^{logn} L={L} {bits}-bit base 2^{base}: {rows}x16 -> {rows * k}x16 ({gb:.2f} GB of digits): fused {res['1']:7.3f} ms  two-step {res['0']:7.3f} ms  ({res['0'] / res['1']:.2f}x)", flush=True)
    del b
