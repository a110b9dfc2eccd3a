"""decompose / mul_decompose at n = 2^16 with 28-bit limbs (the reference's end-to-end default): digits fused into the
head kernel of the forward transform against the two-step path (MXX_HIP_DECOMPOSE_FUSED=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

us = mx.GpuDCRTPolyUniformSampler()
for logn, L, bits, base, shape in ((16, 8, 28, 14, (4, 16)), (16, 8, 24, 12, (4, 16)), (14, 8, 28, 14, (16, 16))):
    n = 1 << logn
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, bits), base)
    ctx = p.ctx()
    b = us.sample_uniform(p, shape[0], shape[1], mx.DistType.FinRingDist())
    k = p.modulus_digits()
    s = us.sample_uniform(p, 2, shape[0] * k, mx.DistType.FinRingDist())
    for fused in ("1", "0"):
        os.environ["MXX_HIP_DECOMPOSE_FUSED"] = fused
        _ffi.reload_env()
        out = b.decompose(); mx.gpu_device_sync()
        td, tm = [], []
        for _ in range(3):
            ctx.timer_start(); out = b.decompose(); td.append(ctx.timer_stop())
            ctx.timer_start(); o2 = s.mul_decompose(b); tm.append(ctx.timer_stop())
        gb = out.row_size() * out.col_size() * L * n * 4 / 1e9
        print(f"n=2^{logn} L={L} {bits}-bit base 2^{base}: {shape[0]}x{shape[1]} -> {out.row_size()}x{out.col_size()} ({gb:.1f} GB of digits)  fused={fused}: decompose {min(td):7.2f} ms  mul_decompose {min(tm):7.2f} ms", flush=True)
        del out, o2
    os.environ.pop("MXX_HIP_DECOMPOSE_FUSED", None)
    del b, s
