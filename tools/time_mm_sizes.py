"""r x k times k x c products at n = 2^14, L = 8: nanoseconds per ring multiply-accumulate for several shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n, L = 16384, 8
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
us = mx.GpuDCRTPolyUniformSampler()
for (r, k, c) in ((32, 32, 32), (32, 64, 16), (64, 64, 64), (64, 256, 64), (128, 64, 128), (32, 1024, 16), (256, 16, 256)):
    a = us.sample_uniform(p, r, k, mx.DistType.FinRingDist())
    b = us.sample_uniform(p, k, c, mx.DistType.FinRingDist())
    out = a * b
    mx.gpu_device_sync()
    best = 1e9
    for _ in range(4):
        ctx.timer_start(); out = a * b; ms = ctx.timer_stop(); best = min(best, ms)
    macs = r * k * c
    gb = (r * k + k * c + r * c) * L * n * 4 / 1e9
    print(f"({r}x{k})*({k}x{c}): {best:8.3f} ms  {best * 1e6 / macs:7.2f} ns per ring-MAC  operands+result {gb:5.1f} GB -> {gb / best:6.2f} TB/s algorithmic", flush=True)
    del a, b, out
