"""Per-launch trace of one M4 chain step (bench.py's M4: n = 256, 12 limbs of 51 bits, base 2^17, d = 2)."""
import sys
sys.path.insert(0, ".")
import mxx_amd as mx
from mxx_amd import _ffi

p = mx.GpuDCRTPolyParams(256, mx.gen_crt_basis(256, 12, 51), 17)
d = 2
sampler = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
td0, a0 = sampler.trapdoor(p, d)
_, a1 = sampler.trapdoor(p, d)
target = a1.slice(0, d, 0, 2 * d)
k = p.modulus_digits()
us = mx.GpuDCRTPolyUniformSampler()
c0 = us.sample_uniform(p, 1, a0.col_size(), mx.DistType.FinRingDist())
bmat = us.sample_uniform(p, d, d * k, mx.DistType.FinRingDist())
mmat = us.sample_uniform(p, d, 3, mx.DistType.FinRingDist())


def step():
    kk = sampler.preimage(p, td0, a0, target)
    c1 = c0 * kk
    md = bmat.mul_decompose(mmat)
    return kk, c1, md


for _ in range(5):
    step()
mx.gpu_device_sync()
_ffi.trace_begin()
step()
rows = _ffi.trace_end()
tot = 0.0
for r in rows:
    tot += r["ms"]
    print(f"{r['ms']*1e3:8.1f} us  {r['bytes']/1e6:8.2f} MB  grid {r['blocks']:>6} x {r['threads']:<5} {r['kernel'][:100]}")
print(f"sum {tot:.3f} ms over {len(rows)} launches")
