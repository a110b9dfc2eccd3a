"""M4 chain step (bench.py's M4): host issue time against device time.  If the loop's issue time ~ its total time the step
is bound by the host mirror's call rate (Python + ctypes), not by the device."""
import sys
import time

sys.path.insert(0, ".")
import mxx_amd as mx

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
for rep in range(3):
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    mx.gpu_device_sync()
    t2 = time.perf_counter()
    print(f"issue {1e3 * (t1 - t0) / n:.4f} ms/step, total {1e3 * (t2 - t0) / n:.4f} ms/step", flush=True)
# one step at a time (device idle at every start): the latency of a step
ts = []
for _ in range(20):
    mx.gpu_device_sync()
    t0 = time.perf_counter()
    step()
    mx.gpu_device_sync()
    ts.append(time.perf_counter() - t0)
print(f"single step, synchronised: median {1e3 * sorted(ts)[10]:.4f} ms", flush=True)
