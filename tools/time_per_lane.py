"""Preimage call time against the samplers' elements per lane (MXX_HIP_SAMPLER_PER_LANE; 0 = the launcher's rule: one
resident round of the chip): the M3A call, its 7-column shard (one rank's share at N = 8) and the M4 ring."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

def run(n, depth, bits, base, d, cols_list, per_lanes):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, depth, bits), base)
    ctx = p.ctx()
    sampler = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, pub = sampler.trapdoor(p, d)
    us = mx.GpuDCRTPolyUniformSampler()
    for cols in cols_list:
        t = us.sample_uniform(p, d, cols, mx.DistType.FinRingDist())
        for pl in per_lanes:
            if pl:
                os.environ["MXX_HIP_SAMPLER_PER_LANE"] = str(pl)
            else:
                os.environ.pop("MXX_HIP_SAMPLER_PER_LANE", None)
            _ffi.reload_env()
            for _ in range(3):
                x = sampler.preimage(p, td, pub, t)
            ts = []
            for _ in range(9):
                ctx.timer_start()
                x = sampler.preimage(p, td, pub, t)
                ts.append(ctx.timer_stop())
            assert pub * x == t
            print(f"n={n} L={depth} cols={cols} per_lane={pl or 'auto'}: call {statistics.median(ts):.4f} ms (min {min(ts):.4f})", flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "m3a"
if which == "shard":
    run(16384, 10, 24, 12, 1, [1, 2, 4, 7, 13, 25], [0])
elif which == "m3a":
    run(16384, 10, 24, 12, 1, [7, 50], [0, 2, 4, 6, 8, 12, 16, 25, 32, 50])
else:
    run(256, 12, 51, 17, 2, [4], [0, 1, 2, 3])
