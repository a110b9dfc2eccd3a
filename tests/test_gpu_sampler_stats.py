"""GPU: statistical acceptance of the samplers at device scale, INDEPENDENT of the CPU restatement.

The bit-exact sampler tests compare the HIP kernels with oracle/oracle_sampling.c, whose keying was designed alongside
them: equality proves determinism and portability, not that the distribution is the reference's D_{Z,sigma,c}
(cuda/src/matrix/MatrixSampling.cu:30-147 - Karney's exact sampler; src/sampler/gpu.rs:363-400 and
src/sampler/trapdoor/gpu.rs:690-811 are the reference's own, much weaker, distribution predicates).  Here >= 10^8 integers
per case are drawn on the device and chi-square-tested against the EXACT discrete Gaussian, computed in this file with
mpmath / float64 - nothing under oracle/ is imported.

  * `sample_distribution(GAUSS)` at sigma = 0.35, 4.578, 8191.5 and the sigma_large of the M3A preimage (1.17e8, wider
    than a 24-bit limb: samples rebuilt from two limbs);
  * Karney with non-zero centres 0.25, 0.5, -0.37 (and 0.74) through the p1 sampler with a diagonal covariance;
  * the G-lattice sampler for base 2^12 / 24-bit limbs (2 digits) and base 2^17 / 51-bit limbs (3 digits): every sample
    lies in the coset {x : sum b^d x_d = v mod q}, and the JOINT histogram over the coset's points follows the spherical
    discrete Gaussian D_{Lambda_v(g), c}, c = (b + 1) sigma (Genise-Micciancio 2018, Alg. 3: what
    cuda/src/matrix/MatrixTrapdoor.cu:701-833 implements);
  * the p1 sampler's empirical covariance against Sigma = [[s^2 - c^2 A, -c^2 B], [-c^2 B, s^2 - c^2 D]];
  * detmath.h's log / cos(2 pi u) evaluated ON THE DEVICE against extended precision, 10^7 points: log and
    sqrt(-2 log) <= 2 ulp, cos(2 pi u) <= 3 ulp.

Fixed seeds: the outcomes are deterministic.  A correct sampler's p-value is uniform on (0, 1); the threshold below is
P_MIN, i.e. a false alarm would have had probability 10^-4 per case when the seeds were chosen.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P_MIN = 1e-4
N_RING = 16384


def seed(gpu, salt):
    return gpu.GpuRngSeed.from_bytes(bytes((13 * i + 7 * salt + 1) & 0xFF for i in range(32)))


def params24(gpu):
    return gpu.GpuDCRTPolyParams(N_RING, gpu.gen_crt_basis(N_RING, 2, 24), 12)


def chi2_pvalue(observed, expected):
    from scipy.stats import chi2

    observed, expected = np.asarray(observed, dtype=np.float64), np.asarray(expected, dtype=np.float64)
    assert abs(observed.sum() - expected.sum()) < 1e-6 * observed.sum()
    stat = float(((observed - expected) ** 2 / expected).sum())
    return stat, len(observed) - 1, float(chi2.sf(stat, len(observed) - 1))


def merge_small_cells(observed, expected, floor=8.0):
    """cells with an expected count below `floor` are pooled into one"""
    small = expected < floor
    if not small.any():
        return observed, expected
    return (np.concatenate([observed[~small], [observed[small].sum()]]),
            np.concatenate([expected[~small], [expected[small].sum()]]))


def centred_two_limbs(r, q0, q1):
    """integers in (-q0 q1 / 2, q0 q1 / 2] from their residues modulo q0 and q1 (Garner)"""
    r0, r1 = r[..., 0, :].astype(np.int64), r[..., 1, :].astype(np.int64)
    t = ((r1 - r0) % q1) * pow(q0, -1, q1) % q1
    v = r0 + q0 * t
    return np.where(v > (q0 * q1) // 2, v - q0 * q1, v)


def exact_pmf(lo, hi, sigma, centre=0.0):
    """D_{Z, sigma, centre} on [lo, hi] with mpmath (50 digits), normalised over a range that holds all but e^-200 of the mass"""
    import mpmath as mp

    mp.mp.dps = 50
    span = int(math.ceil(30 * sigma)) + 5
    w = lambda x: mp.exp(-((x - centre) ** 2) / (2 * mp.mpf(sigma) ** 2))
    total = mp.fsum(w(x) for x in range(int(math.floor(centre)) - span, int(math.ceil(centre)) + span + 1))
    return np.array([float(w(x) / total) for x in range(lo, hi + 1)])


def integer_histogram_test(x, sigma, centre, label):
    """chi-square of integer samples against the exact pmf, one cell per integer, tails pooled"""
    n = x.size
    reach = int(math.ceil(12 * sigma)) + 3
    lo, hi = int(math.floor(centre)) - reach, int(math.ceil(centre)) + reach
    assert x.min() >= lo and x.max() <= hi, (label, int(x.min()), int(x.max()))
    counts = np.bincount((x - lo).ravel(), minlength=hi - lo + 1).astype(np.float64)
    obs, exp = merge_small_cells(counts, exact_pmf(lo, hi, sigma, centre) * n)
    stat, dof, p = chi2_pvalue(obs, exp)
    print(f"{label}: N = {n}, {dof + 1} cells, chi2 = {stat:.1f}, p = {p:.4f}, mean {x.mean():+.5f}, std {x.std():.5f}")
    assert p >= P_MIN, (label, stat, dof, p)
    return p


@pytest.mark.parametrize("sigma", [0.35, 4.578])
def test_gauss_matrix_narrow_widths_against_the_exact_pmf(gpu, sigma):
    """10^8 samples of gpu_matrix_sample_distribution(GAUSS), one chi-square cell per integer"""
    from mxx_amd import _ffi

    p = params24(gpu)
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 218, 28, _ffi.GPU_MATRIX_DIST_GAUSS, sigma, seed(gpu, 1))
    r = m.to_coeff_rns()
    del m
    q0, q1 = p.moduli()
    x = centred_two_limbs(r, q0, q1)
    del r
    assert x.size >= 10**8
    integer_histogram_test(x, sigma, 0.0, f"gauss matrix sigma={sigma}")


@pytest.mark.parametrize("which", ["8191.5", "m3a_sigma_large"])
def test_gauss_matrix_wide_widths_bins_and_residues(gpu, which):
    """Wide Gaussians (the second wider than the 24-bit limbs: integers rebuilt from two limbs): 256 equal-probability
    bins of the exact distribution (for sigma >= 8 the sum of the pmf over a run of integers equals the normal integral
    over it to better than e^-600: Poisson summation), and the fine structure - x mod 64 and x mod 101 uniform, which is
    what Karney's offset draw `j` uniform in [0, ceil(sigma)) must deliver."""
    from scipy.special import ndtr, ndtri

    from mxx_amd import _ffi

    p = params24(gpu)
    if which == "8191.5":
        sigma = 8191.5
    else:  # src/sampler/trapdoor/gpu.rs:15-27,246-249 at the bench_preimage shape (n = 2^14, L = 10, base 2^12, d = 1)
        b, s0, k = 4096.0, 4.578, 20
        s = 1.8 * (b + 1) * s0 * s0 * (math.sqrt(N_RING * k) + math.sqrt(2 * N_RING) + 4.7)
        sigma = math.sqrt(s * s - ((b + 1) * s0) ** 2)
        assert 1.0e8 < sigma < 1.3e8
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 218, 28, _ffi.GPU_MATRIX_DIST_GAUSS, sigma, seed(gpu, 2))
    r = m.to_coeff_rns()
    del m
    q0, q1 = p.moduli()
    x = centred_two_limbs(r, q0, q1).ravel()
    del r
    n = x.size
    assert n >= 10**8 and np.abs(x).max() < 7.5 * sigma
    cells = 256
    edges = np.floor(ndtri(np.arange(1, cells) / cells) * sigma)  # integer edges: a cell is (e_i, e_{i+1}]
    probs = np.diff(np.concatenate([[0.0], ndtr((edges + 0.5) / sigma), [1.0]]))
    counts = np.bincount(np.searchsorted(edges, x, side="left"), minlength=cells).astype(np.float64)
    stat, dof, pv = chi2_pvalue(counts, probs * n)
    print(f"gauss matrix sigma={sigma:.6g}: {cells} bins chi2 = {stat:.1f}, p = {pv:.4f}, std / sigma = {x.std() / sigma:.6f}")
    assert pv >= P_MIN
    assert abs(x.std() / sigma - 1.0) < 5.0 / math.sqrt(2 * n)  # the estimator's own 5 sigma
    assert abs(x.mean()) < 5.0 * sigma / math.sqrt(n)
    for mod in (64, 101):
        c = np.bincount(np.mod(x, mod), minlength=mod).astype(np.float64)
        stat, dof, pv = chi2_pvalue(c, np.full(mod, n / mod))
        print(f"    x mod {mod}: chi2 = {stat:.1f} / {dof}, p = {pv:.4f}")
        assert pv >= P_MIN


@pytest.mark.parametrize("ratio,tp2_rows", [(0.25, (-1, -2)), (0.37, (1, -2))])
def test_karney_nonzero_centres_through_the_p1_sampler(gpu, ratio, tp2_rows):
    """gpu_matrix_sample_p1_full_cached with A = B = D = 0: Sigma = s^2 I, so both coordinates are independent
    D_{Z, s, mu_t} with mu_t = -c^2 / (s^2 - c^2) * tp2_t - Karney's sampler with centres 0.25 and 0.5
    (ratio 1/4, tp2 = -1, -2), then -0.37 and 0.74.  10^8 integers per centre, one chi-square cell per integer."""
    p = params24(gpu)
    q = p.moduli()
    n, cols = N_RING, 6104
    s_par = 3.3
    c_par = math.sqrt(ratio * s_par * s_par / (1.0 + ratio))  # c^2 / (s^2 - c^2) = ratio
    M = gpu.GpuDCRTPolyMatrix
    zero = M.zero(p, 1, 1)
    zero.intt_all_in_place()
    cache = M.create_p1_covariance_cache(zero, zero, zero, c_par, s_par, 4.578)
    tp2 = np.zeros((2, cols, 2, n), dtype=np.uint64)
    for row, v in enumerate(tp2_rows):
        for l in range(2):
            tp2[row, :, l, :] = v % q[l]
    out = M.sample_p1_full_cached(cache, M.from_rns(p, tp2, False), seed(gpu, 3))
    del tp2
    r = out.to_coeff_rns()
    del out
    x = centred_two_limbs(r, q[0], q[1])
    del r
    for row, v in enumerate(tp2_rows):
        mu = -ratio * v
        assert x[row].size >= 10**8
        integer_histogram_test(x[row], s_par, mu, f"p1 / Karney centre {mu:+.2f}, s = {s_par}")
    # the two coordinates are independent (Sigma is diagonal)
    a, b = x[0].ravel().astype(np.float64), x[1].ravel().astype(np.float64)
    corr = float(np.corrcoef(a[:20_000_000], b[:20_000_000])[0, 1])
    assert abs(corr) < 5.0 / math.sqrt(20_000_000), corr


def _adjugate_int(B):
    """integer adjugate and determinant of a small integer matrix (exact, Python fractions)"""
    from fractions import Fraction

    k = B.shape[0]
    A = [[Fraction(int(B[i, j])) for j in range(k)] + [Fraction(int(i == j)) for j in range(k)] for i in range(k)]
    det = Fraction(1)
    for col in range(k):
        piv = next(r for r in range(col, k) if A[r][col] != 0)
        if piv != col:
            A[col], A[piv] = A[piv], A[col]
            det = -det
        det *= A[col][col]
        pv = A[col][col]
        A[col] = [v / pv for v in A[col]]
        for r in range(k):
            if r != col and A[r][col] != 0:
                f = A[r][col]
                A[r] = [a - f * b for a, b in zip(A[r], A[col])]
    adj = np.array([[int(A[i][k + j] * det) for j in range(k)] for i in range(k)], dtype=np.int64)
    return adj, int(det)


def coset_lattice(base, modulus, dpt, v):
    """Genise-Micciancio's basis of Lambda(g) = {x in Z^dpt : sum base^d x_d = 0 (mod modulus)} - columns
    base e_d - e_(d+1) and the modulus' digits - and the coset shift t = the digits of v"""
    B = np.zeros((dpt, dpt), dtype=np.int64)
    for d in range(dpt - 1):
        B[d, d], B[d + 1, d] = base, -1
    B[:, dpt - 1] = [(modulus // base**d) % base for d in range(dpt)]
    t = np.array([(v // base**d) % base for d in range(dpt)], dtype=np.int64)
    adj, det = _adjugate_int(B)
    assert abs(det) == modulus
    return B, t, adj, det


def coset_points(B, t, c, reach):
    """y coordinates and probabilities of every point x = t + B y with |x| <= reach * c under the spherical discrete
    Gaussian of width c on the coset (float64; the mass beyond reach = 7.5 is below 1e-12)"""
    dpt = B.shape[0]
    Binv = np.linalg.inv(B.astype(np.float64))
    R = reach * c
    y0 = -Binv @ t.astype(np.float64)
    half = [int(math.ceil(R * np.linalg.norm(Binv[i]))) + 1 for i in range(dpt)]
    axes = [np.arange(int(round(y0[i])) - half[i], int(round(y0[i])) + half[i] + 1, dtype=np.int64) for i in range(dpt)]
    grids = np.meshgrid(*axes, indexing="ij")
    Y = np.stack([g.ravel() for g in grids], axis=0)
    X = t[:, None] + B @ Y
    n2 = (X.astype(np.float64) ** 2).sum(axis=0)
    keep = n2 <= R * R
    w = np.exp(-n2[keep] / (2.0 * c * c))
    return Y[:, keep], w / w.sum(), [(int(a[0]), len(a)) for a in axes]


@pytest.mark.parametrize("bits,base_bits,value", [(24, 12, 0x5A5A5A), (24, 12, 1), (51, 17, 0x2F0F0F0F0F0F1)])
def test_gauss_samp_gq_joint_distribution_over_the_coset(gpu, bits, base_bits, value):
    """>= 10^8 digits of gpu_matrix_gauss_samp_gq_arb_base on a constant input (base 2^12 with 24-bit limbs: 2 digits per
    tower; base 2^17 with 51-bit limbs: 3).  Per tower: every sample is a point of the coset Lambda_v(g) (x - t = B y
    with y integral - an exact check on each sample), and the JOINT histogram over the coset's points within 7.5 c
    matches the spherical discrete Gaussian D_{Lambda_v(g), c}, c = (b + 1) sigma; every digit's marginal follows."""
    n = N_RING
    moduli = gpu.gen_crt_basis(n, 2, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, base_bits)
    base, sigma = 1 << base_bits, 4.578
    c = (base + 1) * sigma
    dpt = -(-bits // base_bits)
    cols = 1526 if dpt == 2 else 1018  # 2 towers x dpt digits x cols x 2^14 >= 10^8 integers
    vs = [(value * (3 + 2 * l)) % moduli[l] for l in range(2)]
    src = np.zeros((1, cols, 2, n), dtype=np.uint64)
    for l in range(2):
        src[0, :, l, :] = vs[l]
    z = gpu.GpuDCRTPolyMatrix.from_rns(p, src, False).gauss_samp_gq_arb_base(c, sigma, seed(gpu, 4 + bits + (value & 1)), coeff_out=True)
    del src
    assert z.size() == (2 * dpt, cols) and not z.is_ntt
    zr = z.to_rns()  # COEFF residues as they stand
    del z
    q0 = moduli[0]
    total = 0
    for tower in range(2):
        lim = zr[tower * dpt:(tower + 1) * dpt, :, 0, :].astype(np.int64).reshape(dpt, -1)  # limb 0 holds every digit as a small integer
        x = np.where(lim > q0 // 2, lim - q0, lim)
        other = zr[tower * dpt:(tower + 1) * dpt, :, 1, :].astype(np.int64).reshape(dpt, -1)
        assert np.array_equal(np.where(other > moduli[1] // 2, other - moduli[1], other), x)  # the same integer in both limbs
        del lim, other
        total += x.size
        B, t, adj, det = coset_lattice(base, moduli[tower], dpt, vs[tower])
        u = x - t[:, None]
        assert np.abs(u).max() < 1 << 25
        w = adj @ u  # |adj| <= base^(dpt-1) = 2^34, |u| < 2^25, dpt terms: below 2^61
        assert not np.any(w % det), "a sample left the coset Lambda_v(g)"
        y = w // det
        del u, w
        Y, prob, axes = coset_points(B, t, c, 7.5)
        strides, size = [], 1
        for lo, ln in axes:
            strides.append(size)
            size *= ln
        inside = np.ones(y.shape[1], dtype=bool)
        key = np.zeros(y.shape[1], dtype=np.int64)
        for i, (lo, ln) in enumerate(axes):
            inside &= (y[i] >= lo) & (y[i] < lo + ln)
            key += (y[i] - lo) * strides[i]
        counts_all = np.bincount(key[inside], minlength=size).astype(np.float64)
        pkey = sum((Y[i] - axes[i][0]) * strides[i] for i in range(dpt))
        observed = counts_all[pkey]
        stray = float(x.shape[1] - observed.sum())  # beyond 7.5 c, or box corners outside the ball: pooled with the small cells
        nsamp = x.shape[1]
        expected = prob * nsamp
        small = expected < 8.0
        obs = np.concatenate([observed[~small], [observed[small].sum() + stray]])
        exp = np.concatenate([expected[~small], [expected[small].sum()]])
        stat, dof, pv = chi2_pvalue(obs, exp)
        print(f"G-sampler {bits}-bit / base 2^{base_bits}, tower {tower}, v = {vs[tower]}: {nsamp} samples over {len(prob)} coset "
              f"points ({dof + 1} cells), chi2 = {stat:.1f}, p = {pv:.4f}; digit std / c = {[round(float(x[d].std() / c), 5) for d in range(dpt)]}")
        assert pv >= P_MIN, (tower, stat, dof, pv)
    assert total >= 10**8


def test_p1_sampler_covariance_matches_sigma(gpu):
    """gpu_matrix_sample_p1_full_cached with constant polynomials A, B, D (every coefficient index then shares one 2 x 2
    Sigma = [[s^2 - c^2 A, -c^2 B], [-c^2 B, s^2 - c^2 D]]) and tp2 = 0: the empirical covariance of 5 x 10^7
    sample pairs lies within 4 standard deviations of the estimator of Sigma, entry by entry (Var[cov_ij] = (S_ii S_jj + S_ij^2) / N)."""
    p = params24(gpu)
    q = p.moduli()
    n, cols = N_RING, 3052
    c_par, s_par = 6.0, 40.0
    A, Bv, D = 17, -9, 23
    M = gpu.GpuDCRTPolyMatrix

    def const(v):
        a = np.zeros((1, 1, 2, n), dtype=np.uint64)
        for l in range(2):
            a[0, 0, l, :] = v % q[l]
        return M.from_rns(p, a, False)

    cache = M.create_p1_covariance_cache(const(A), const(Bv), const(D), c_par, s_par, 4.578)
    out = M.sample_p1_full_cached(cache, M.from_rns(p, np.zeros((2, cols, 2, n), dtype=np.uint64), False), seed(gpu, 6))
    x = centred_two_limbs(out.to_coeff_rns(), q[0], q[1]).reshape(2, -1).astype(np.float64)
    del out
    N = x.shape[1]
    S = np.array([[s_par**2 - c_par**2 * A, -c_par**2 * Bv], [-c_par**2 * Bv, s_par**2 - c_par**2 * D]])
    emp = (x @ x.T) / N
    print("Sigma", S.tolist(), "empirical", emp.tolist(), "mean", x.mean(axis=1).tolist())
    for i in range(2):
        assert abs(x[i].mean()) < 5.0 * math.sqrt(S[i, i] / N)
        for j in range(2):
            sd = math.sqrt((S[i, i] * S[j, j] + S[i, j] ** 2) / N)
            assert abs(emp[i, j] - S[i, j]) < 4.0 * sd, (i, j, emp[i, j], S[i, j], sd)


def test_detmath_on_the_device_against_extended_precision(gpu):
    """mxx_amd/csrc/detmath.h evaluated where it runs: 10^7 arguments (uniform in (0, 1), plus small ones down to 2^-53 and
    the quadrant boundaries) through gpupoly_detmath_eval, against numpy's 80-bit long double: det_log and
    sqrt(-2 det_log) within 2 ulp (measured 0.84 / 0.96); det_cos2pi within 3 ulp of the result (measured 2.35) where
    |cos| >= 2^-10 and 2^-52 absolute near its zeros (the argument reduction 4u - round(4u) is exact, so the error there is the final angle's rounding)."""
    import ctypes as C

    from mxx_amd import _ffi

    p = params24(gpu)
    rng = np.random.default_rng(11)
    n = 10_000_000
    u = rng.random(n)
    u[: n // 10] *= 2.0 ** -rng.integers(1, 52, n // 10)
    u[:7] = [2.0**-53, 1 - 2.0**-53, 0.5, 0.25, 0.75, 0.125, 1 / 3]
    u = np.clip(u, 2.0**-53, 1 - 2.0**-53)
    dp = C.POINTER(C.c_double)

    def device(fn):
        out = np.empty(n, dtype=np.float64)
        _ffi.check_status(_ffi.lib().gpupoly_detmath_eval(p.ctx_raw(), fn, u.ctypes.data_as(dp), out.ctypes.data_as(dp), n), "gpupoly_detmath_eval")
        return out

    ul = u.astype(np.longdouble)
    assert np.finfo(np.longdouble).nmant >= 63
    want_log = np.log(ul)
    got = device(0)
    ulp = np.abs(np.spacing(want_log.astype(np.float64)))
    worst = float(np.max(np.abs(got.astype(np.longdouble) - want_log) / ulp))
    print(f"det_log on the device: worst error {worst:.3f} ulp over {n} points")
    assert worst <= 2.0
    want_r = np.sqrt(-2 * want_log)
    got = device(2)
    worst = float(np.max(np.abs(got.astype(np.longdouble) - want_r) / np.abs(np.spacing(want_r.astype(np.float64)))))
    print(f"sqrt(-2 det_log): worst error {worst:.3f} ulp")
    assert worst <= 2.0
    two_pi = 2 * np.longdouble("3.14159265358979323846264338327950288")
    want_cos = np.cos(two_pi * ul)
    got = device(1)
    err = np.abs(got.astype(np.longdouble) - want_cos)
    big = np.abs(want_cos) >= 2.0**-10
    worst = float(np.max(err[big] / np.abs(np.spacing(want_cos[big].astype(np.float64)))))
    print(f"det_cos2pi: worst error {worst:.3f} ulp where |cos| >= 2^-10; {float(err[~big].max()):.3e} absolute near the zeros")
    # measured on gfx950: 2.35 ulp (the final angle d * (pi / 2) carries the rounding of the constant and of the product,
    # and results just above a power of two halve the ulp); log and sqrt(-2 log) stay below 1 ulp
    assert worst <= 3.0 and float(err[~big].max()) <= 2.0**-52
    assert got[2] == -1.0 and abs(got[3]) < 2.0**-52 and abs(got[4]) < 2.0**-52
