"""Per-launch trace of one preimage call at the M3A shape (library launch trace: kernel, grid, algorithmic bytes, ms)."""
import sys
sys.path.insert(0, ".")
import mxx_amd as mx
from mxx_amd import _ffi

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 50
L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = mx.GpuDCRTPolyParams(16384, mx.gen_crt_basis(16384, L, 24), 12)
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
td, pub = s.trapdoor(p, 1)
t = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, cols, mx.DistType.FinRingDist())
for _ in range(3):
    x = s.preimage(p, td, pub, t)
mx.gpu_device_sync()
_ffi.trace_begin()
x = s.preimage(p, td, pub, t)
rows = _ffi.trace_end()
tot = 0.0
for r in rows:
    name, blocks, threads, nbytes, ms = r['kernel'], r['blocks'], r['threads'], r['bytes'], r['ms']
    tot += ms
    gbs = nbytes / ms / 1e6 if ms > 0 and nbytes else 0.0
    print(f"{ms*1e3:9.1f} us  {gbs:8.0f} GB/s  {nbytes/1e6:9.1f} MB  grid {blocks:>8} x {threads:<5} {name[:90]}")
print(f"sum {tot:.3f} ms over {len(rows)} launches")
