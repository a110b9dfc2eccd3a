"""One-off costs around the preimage path at M3A parameters: trapdoor generation, covariance cache, serialisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n, L = 16384, 10
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx = p.ctx()
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
def wall(fn, reps=3):
    fn(); mx.gpu_device_sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); mx.gpu_device_sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
    return best, r
for d in (1, 2, 4):
    ms, (td, pub) = wall(lambda: s.trapdoor(p, d))
    print(f"d={d}: trapdoor generation (R, E, A; products cached) {ms:8.3f} ms wall")
    us = mx.GpuDCRTPolyUniformSampler()
    tg = us.sample_uniform(p, d, 8, mx.DistType.FinRingDist())
    ms1, _ = wall(lambda: s.preimage(p, td, pub, tg), 1)   # first call builds the covariance cache
    td2, pub2 = s.trapdoor(p, d)
    t0 = time.perf_counter(); s.preimage(p, td2, pub2, tg); mx.gpu_device_sync(); first = (time.perf_counter() - t0) * 1e3
    ms2, _ = wall(lambda: s.preimage(p, td2, pub2, tg), 3)
    print(f"d={d}: preimage of 8 columns: first call {first:8.3f} ms (builds the p1 covariance cache), then {ms2:8.3f} ms")
    ms, blob = wall(lambda: td.to_compact_bytes(), 2)
    print(f"d={d}: trapdoor to_compact_bytes {ms:8.3f} ms ({len(blob) / 1e6:.1f} MB)")
    ms, _ = wall(lambda: mx.GpuDCRTTrapdoor.from_compact_bytes(p, blob), 2)
    print(f"d={d}: trapdoor from_compact_bytes {ms:8.3f} ms")
