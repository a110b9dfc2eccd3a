"""Stress primitives at scale to locate flaky behaviour."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mxx_amd as mx

n, depth = 16384, 10
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, depth, 24), 12)
us = mx.GpuDCRTPolyUniformSampler()
rows, cols = 20, 50
bad = {"clone_eq": 0, "ntt_rt_eq": 0, "ntt_rt_host": 0, "fwd_det": 0}
for it in range(10):
    z = us.sample_uniform(p, rows, cols, mx.DistType.FinRingDist())
    c = z.clone()
    if not (c == z): bad["clone_eq"] += 1
    host0 = z.to_rns()
    zc = z.clone(); zc.intt_all_in_place(); zc.ntt_all_in_place()
    if not (zc == z): bad["ntt_rt_eq"] += 1
    if not np.array_equal(zc.to_rns(), host0): bad["ntt_rt_host"] += 1
    a = z.clone(); a.intt_all_in_place(); h1 = a.to_rns()
    b = z.clone(); b.intt_all_in_place(); h2 = b.to_rns()
    if not np.array_equal(h1, h2):
        bad["fwd_det"] += 1
        diff = np.argwhere(h1 != h2)
        print("intt nondeterministic: ndiff", len(diff), "first", diff[:3].tolist(), flush=True)
print(bad, flush=True)
