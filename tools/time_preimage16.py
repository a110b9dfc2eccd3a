"""Preimage calls on the reference's end-to-end ring: n = 2^16, 28-bit limbs (L = 6), base 2^14, d = 1, 16 target columns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n, L = 1 << 16, 6
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 28), 14)
ctx = p.ctx()
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
td, pub = s.trapdoor(p, 1)
tg = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, 16, mx.DistType.FinRingDist())
x = s.preimage(p, td, pub, tg)
assert pub * x == tg
mx.gpu_device_sync()
ts = []
for _ in range(8):
    ctx.timer_start(); x = s.preimage(p, td, pub, tg); ts.append(ctx.timer_stop())
print(f"n=2^16, L={L} (28-bit), base 2^14, k={p.modulus_digits()}: preimage of 16 columns {min(ts):.3f} ms (median {sorted(ts)[len(ts)//2]:.3f})")
