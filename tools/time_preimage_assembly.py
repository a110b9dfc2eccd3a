"""Preimage call under the two assemblies of the result (mxx_amd/trapdoor.py: TRAFFIC_BOUND_BYTES) at rings with and
without the fused NTT + add kernel:  python tools/time_preimage_assembly.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
import mxx_amd.trapdoor as T

for logn, bits, depth, base, cols in ((14, 24, 10, 12, 50), (16, 28, 6, 14, 8), (15, 24, 8, 12, 16), (14, 51, 4, 17, 16)):
    n = 1 << logn
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, depth, bits), base)
    ctx = p.ctx()
    sampler = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, A = sampler.trapdoor(p, 1)
    target = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, cols, mx.DistType.FinRingDist())
    line = f"n=2^{logn} {bits}-bit x {depth}, {cols} columns:"
    for name, bound in (("small-operand assembly", 1 << 62), ("large-operand assembly", 0)):
        T.TRAFFIC_BOUND_BYTES = bound
        for _ in range(2):
            x = sampler.preimage(p, td, A, target)
        ts = []
        for _ in range(5):
            ctx.timer_start()
            x = sampler.preimage(p, td, A, target)
            ts.append(ctx.timer_stop())
        assert A * x == target
        line += f"  {name} {sorted(ts)[2]:.3f} ms"
    print(line, flush=True)
