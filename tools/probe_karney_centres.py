"""Exploration: per-integer residuals of the p1 / Karney sampler with non-zero centres, pooled over several seeds."""
import math, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mxx_amd as gpu
import test_gpu_sampler_stats as st

p = st.params24(gpu)
q = p.moduli()
n, cols = st.N_RING, 6104
s_par = float(sys.argv[1]) if len(sys.argv) > 1 else 3.3
ratio = 0.25
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
c_par = math.sqrt(ratio * s_par * s_par / (1.0 + ratio))
M = gpu.GpuDCRTPolyMatrix
zero = M.zero(p, 1, 1); zero.intt_all_in_place()
cache = M.create_p1_covariance_cache(zero, zero, zero, c_par, s_par, 4.578)
rows = (-1, -2)
reach = int(math.ceil(12 * s_par)) + 3
tot = {}
for sd in range(seeds):
    tp2 = np.zeros((2, cols, 2, n), dtype=np.uint64)
    for row, v in enumerate(rows):
        for l in range(2):
            tp2[row, :, l, :] = v % q[l]
    out = M.sample_p1_full_cached(cache, M.from_rns(p, tp2, False), st.seed(gpu, 100 + sd))
    x = st.centred_two_limbs(out.to_coeff_rns(), q[0], q[1])
    for row, v in enumerate(rows):
        mu = -ratio * v
        lo = int(math.floor(mu)) - reach
        c = np.bincount((x[row] - lo).ravel(), minlength=2 * reach + 3).astype(np.float64)
        tot[row] = tot.get(row, 0) + c
        pm = st.exact_pmf(lo, lo + len(c) - 1, s_par, mu) * x[row].size
        o, e = st.merge_small_cells(c, pm)
        print("seed", sd, "centre", mu, "chi2", st.chi2_pvalue(o, e))
for row, v in enumerate(rows):
    mu = -ratio * v
    lo = int(math.floor(mu)) - reach
    c = tot[row]
    N = c.sum()
    pm = st.exact_pmf(lo, lo + len(c) - 1, s_par, mu) * N
    o, e = st.merge_small_cells(c, pm)
    print("POOLED centre", mu, "N", N, st.chi2_pvalue(o, e))
    for i in range(len(c)):
        if pm[i] > 1000:
            print(f"   x={lo + i:4d} obs {c[i]:12.0f} exp {pm[i]:14.1f} z {(c[i] - pm[i]) / math.sqrt(pm[i]):+6.2f} rel {(c[i] - pm[i]) / pm[i]:+.2e}")
