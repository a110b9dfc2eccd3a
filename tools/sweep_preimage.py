"""Preimage call time against the number of target columns (what a rank sees under column sharding), M3A parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n, L = 16384, 10
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
td, pub = s.trapdoor(p, 1)
us = mx.GpuDCRTPolyUniformSampler()
for cols in (50, 25, 13, 7, 6, 2, 1):
    t = us.sample_uniform(p, 1, cols, mx.DistType.FinRingDist())
    x = s.preimage(p, td, pub, t)
    mx.gpu_device_sync()
    best = 1e9
    for _ in range(4):
        ctx.timer_start(); x = s.preimage(p, td, pub, t); ms = ctx.timer_stop(); best = min(best, ms)
    print(f"{cols:3d} target columns: {best:7.3f} ms  ({best / cols:6.3f} ms per column, {cols / best * 1e3:7.0f} preimages/s)", flush=True)
