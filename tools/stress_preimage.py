"""Stress: repeated preimages at the bench shape, checking every intermediate exact predicate."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd.sampler import random_gpu_rng_seed

n, depth, base = 16384, int(os.environ.get("DEPTH", "10")), 12
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, depth, 24), base)
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
td, A = s.trapdoor(p, 1)
G = mx.GpuDCRTPolyMatrix.gadget_matrix(p, 1)
us = mx.GpuDCRTPolyUniformSampler()
bad = 0
for it in range(reps):
    target = us.sample_uniform(p, 1, 50, mx.DistType.FinRingDist())
    # G-sampler relation alone
    keep = target.clone()
    z = target.clone().gauss_samp_gq_arb_base(s.c, s.sigma, random_gpu_rng_seed())
    ok_g = (G * z == keep)
    # NTT round trip of z
    zc = z.clone(); zc.intt_all_in_place(); zc.ntt_all_in_place()
    ok_ntt = (zc == z)
    x = s.preimage(p, td, A, target)
    ok_x = (A * x == target)
    if not (ok_g and ok_ntt and ok_x):
        bad += 1
        print(f"iter {it}: G*z==v {ok_g}  ntt-roundtrip {ok_ntt}  A*x==u {ok_x}", flush=True)
print(f"done: {bad} bad of {reps}", flush=True)
