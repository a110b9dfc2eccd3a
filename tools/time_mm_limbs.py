"""64x64 * 64x64 product at n = 2^14 for several limb counts: ms per limb (a power-of-two polynomial stride - L = 8 -
puts every panel row of a workgroup on the same few L2 / HBM channels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n = 16384
us = mx.GpuDCRTPolyUniformSampler()
for L in (6, 7, 8, 9, 10, 16):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
    ctx = p.ctx()
    a = us.sample_uniform(p, 64, 64, mx.DistType.FinRingDist())
    b = us.sample_uniform(p, 64, 64, mx.DistType.FinRingDist())
    c = a * b
    mx.gpu_device_sync()
    best = 1e9
    for _ in range(5):
        ctx.timer_start(); c = a * b; ms = ctx.timer_stop(); best = min(best, ms)
    print(f"L={L:2d}: {best:.3f} ms  -> {best / L:.4f} ms per limb", flush=True)
    del a, b, c
