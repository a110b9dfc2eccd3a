import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
n, L = 16384, 8
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 51), 17)
ctx = p.ctx(); us = mx.GpuDCRTPolyUniformSampler()
for (r, k, c) in ((1, 30, 120), (1, 30, 15), (2, 20, 50), (4, 64, 64), (8, 256, 64), (1, 256, 8)):
    a = us.sample_uniform(p, r, k, mx.DistType.FinRingDist()); b = us.sample_uniform(p, k, c, mx.DistType.FinRingDist())
    out = a * b; mx.gpu_device_sync(); best = 1e9
    for _ in range(4):
        ctx.timer_start(); out = a * b; best = min(best, ctx.timer_stop())
    gb = (r * k + k * c + r * c) * L * n * 8 / 1e9
    print(f"u64 ({r}x{k})*({k}x{c}): {best*1e3:8.1f} us  {gb/best:6.2f} TB/s algorithmic", flush=True)
    del a, b, out
