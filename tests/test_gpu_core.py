"""GPU parity: deterministic ring-matrix ops through the C ABI vs the CPU oracle.

Bit-exact (integer work).  Shapes follow the reference's own GPU unit tests
(src/poly/dcrt/gpu.rs tests, src/matrix/gpu_dcrt_poly.rs:1946-2720) plus the
synthetic configs of BASELINE.json.
"""
import os

import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu

# (n, depth, bits, base_bits): u32 and u64 word paths, every NTT kernel family
PARAM_SETS = [
    (4, 2, 17, 1),       # DCRTPolyParams::default()
    (16, 3, 18, 6),
    (128, 2, 17, 1),     # reference GPU test params
    (128, 2, 16, 4),
    (128, 2, 16, 8),
    (1024, 3, 24, 12),   # tuned LDS kernel, logN=10
    (4096, 2, 24, 12),   # BASELINE config 0
    (256, 3, 51, 17),    # u64 words, config-5 style
    (1024, 5, 51, 17),   # reference norm-test params (u64, tuned kernel)
]


@pytest.mark.parametrize("n,depth,bits,base", PARAM_SETS)
def test_rns_roundtrip_and_levels(gpu, oracle, n, depth, bits, base):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    x = rand_matrix(oracle, 1, 2, 3, moduli, n)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, True)
    assert np.array_equal(m.to_rns(), x)
    # lower level: only limbs 0..=level
    lvl = depth - 2
    if lvl >= 0:
        xl = x[:, :, : lvl + 1]
        ml = gpu.GpuDCRTPolyMatrix.from_rns(p, xl, False)
        assert ml.level == lvl
        assert np.array_equal(ml.to_rns(), xl)
        ml.ntt_all_in_place()
        assert np.array_equal(ml.to_rns(), oracle.matrix_ntt(xl, moduli[: lvl + 1]))


@pytest.mark.parametrize("n,depth,bits,base", PARAM_SETS)
def test_ntt_intt_vs_oracle(gpu, oracle, n, depth, bits, base):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    x = rand_matrix(oracle, 2, 2, 2, moduli, n)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    ev = m.to_rns()
    assert np.array_equal(ev, oracle.matrix_ntt(x, moduli))
    m.ntt_all_in_place()  # idempotent on the tag
    assert np.array_equal(m.to_rns(), ev)
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
    # worst-case inputs: all q-1, and zero
    top = np.broadcast_to((np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1), x.shape).copy()
    mt = gpu.GpuDCRTPolyMatrix.from_rns(p, top, False)
    mt.ntt_all_in_place()
    assert np.array_equal(mt.to_rns(), oracle.matrix_ntt(top, moduli))


@pytest.mark.parametrize("bits", [17, 30, 51, 61])
@pytest.mark.parametrize("logn", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_ntt_small_rings(gpu, oracle, hip_env, logn, bits):
    """n = 2..512 (the reference's unit-test rings and BASELINE configs[4]'s n = 256) in both word widths, moduli from 17 to
    61 bits, 63 vectors per call, extreme residues, both directions - against the CPU restatement, default dispatch and
    the forced generic kernel.  (A wave-per-vector kernel for these sizes was built in round 3 and measured slower than
    the 128-thread LDS kernel - profiles/r03_notes.md.)  Round 5: 64-bit words with moduli below 2^51 take the
    double-precision form of that kernel (nttf::small_kernel) by default; MXX_HIP_NTT64=int and the forced generic path
    keep the integer one - all three must give the oracle's bits."""
    n = 1 << logn
    depth = 3
    moduli = oracle.gen_crt_basis(n, depth, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 4)
    x = rand_matrix(oracle, 40 + logn, 3, 7, moduli, n)     # 63 vectors: never a whole number of waves' worth below 64 points
    x[0, 0] = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(-1, 1)
    x[0, 1] = 0
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    ev = oracle.matrix_ntt(x, moduli)
    assert np.array_equal(m.to_rns(), ev)
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
    hip_env.set("MXX_HIP_NTT64", "int")
    i64 = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    i64.ntt_all_in_place()
    assert np.array_equal(i64.to_rns(), ev)
    i64.intt_all_in_place()
    assert np.array_equal(i64.to_rns(), x)
    hip_env.unset("MXX_HIP_NTT64")
    hip_env.set("MXX_HIP_NTT_PATH", "generic")
    g = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    g.ntt_all_in_place()
    assert np.array_equal(g.to_rns(), ev)
    g.intt_all_in_place()
    assert np.array_equal(g.to_rns(), x)


@pytest.mark.parametrize("logn,bits", [(ln, b) for ln in (10, 11, 12, 13, 14) for b in (51, 50, 49, 45, 33)] +
                         [(ln, b) for ln in (15, 16, 17) for b in (51, 49, 32)])
def test_ntt_u64_double_precision_kernels(gpu, oracle, hip_env, logn, bits):
    """64-bit words with moduli below 2^51 take the double-precision transforms (ntt_f64.h: residues as exact integers in
    doubles, products through FMA, folds inserted at compile time so that nothing reaches 2^53 - every second stage at 51
    bits, once per pass below 2^49, never inside a pass below 2^40; whole-vector kernels up to 2^14, head / tail + sub-vector
    kernels at 2^15..2^17 - 32-bit limbs at 2^16 are the default of the reference's Montgomery mod-q tests,
    tests/test_gpu_ggh15_montgomery_modq_arith.rs:39-40): bit-exact against the CPU restatement and against the integer kernels
    (MXX_HIP_NTT64=int) on random vectors and on the inputs that maximise every intermediate (all q - 1, alternating
    0 / q - 1 at several periods, a spike), both directions, and the decomposition fused into the forward transform."""
    n = 1 << logn
    moduli = oracle.gen_crt_basis(n, 3, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 17)
    top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1)
    pats = [rand_matrix(oracle, 170 + logn, 1, 2, moduli, n), np.broadcast_to(top, (1, 1, len(moduli), n)).copy()]
    for period in (2, 64, n):
        m = np.broadcast_to(top, (1, 1, len(moduli), n)).copy()
        m[..., (np.arange(n) // (period // 2)) % 2 == 1] = 0
        pats.append(m)
    spike = np.zeros((1, 1, len(moduli), n), dtype=np.uint64)
    spike[..., n - 1] = top[..., 0]
    pats.append(spike)
    x = np.concatenate(pats, axis=1)
    want = oracle.matrix_ntt(x, moduli)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), want)
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
    e = gpu.GpuDCRTPolyMatrix.from_rns(p, x, True)  # the same patterns as evaluation-domain inputs of the inverse
    e.intt_all_in_place()
    assert np.array_equal(e.to_rns(), oracle.matrix_ntt(x, moduli, inverse=True))
    dec = gpu.GpuDCRTPolyMatrix.from_rns(p, x[:, :2], False).decompose()
    want_dec = oracle.matrix_ntt(oracle.decompose(x[:, :2], moduli, 17), moduli)
    assert np.array_equal(dec.to_rns(), want_dec)
    hip_env.set("MXX_HIP_NTT64", "int")
    mi = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    mi.ntt_all_in_place()
    assert np.array_equal(mi.to_rns(), want)
    assert np.array_equal(gpu.GpuDCRTPolyMatrix.from_rns(p, x[:, :2], False).decompose().to_rns(), want_dec)


@pytest.mark.parametrize("path", ["generic", "global"])
@pytest.mark.parametrize("n,depth,bits,base", [(128, 2, 17, 1), (1024, 3, 24, 12), (1024, 5, 51, 17)])
def test_ntt_alternate_kernels(gpu, oracle, hip_env, path, n, depth, bits, base):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    x = rand_matrix(oracle, 3, 1, 3, moduli, n)
    hip_env.set("MXX_HIP_NTT_PATH", path)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), oracle.matrix_ntt(x, moduli))
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)


def test_ntt_n16384_bench_moduli(gpu, oracle):
    """The bench ring: n=2^14, 24-bit limbs; every limb of the 15-limb basis once."""
    n = 16384
    p = make_params(gpu, oracle, n, 15, 24, 12)
    moduli = p.moduli()
    x = rand_matrix(oracle, 4, 1, 2, moduli, n)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), oracle.matrix_ntt(x, moduli))
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)


def test_ntt_n16384_both_kernel_designs(gpu, oracle, hip_env):
    """grouped with signed butterflies (ntt14.h, default), grouped unsigned, and whole-vector-in-LDS
    (ntt_lds.h) 2^14 kernels give identical bits."""
    n = 16384
    p = make_params(gpu, oracle, n, 4, 24, 12)
    moduli = p.moduli()
    x = rand_matrix(oracle, 6, 3, 3, moduli, n)
    want = oracle.matrix_ntt(x, moduli)
    for design in ("grouped", "unsigned", "whole"):
        hip_env.set("MXX_HIP_NTT14", design)
        m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
        m.ntt_all_in_place()
        assert np.array_equal(m.to_rns(), want), design
        m.intt_all_in_place()
        assert np.array_equal(m.to_rns(), x), design


@pytest.mark.parametrize("bits", [24, 22, 25])
def test_ntt_n16384_extreme_inputs(gpu, oracle, bits):
    """Redundant-form bounds of the lazy butterflies: all residues q-1, alternating 0 / q-1 in every
    period, a single spike - for the largest 24-bit primes (signed form), 22-bit primes, and 25-bit primes
    (past the signed form's bound: the unsigned grouped kernels take over)."""
    n = 16384
    moduli = oracle.gen_crt_basis(n, 3, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 12)
    top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1)
    pats = [np.broadcast_to(top, (1, 1, len(moduli), n)).copy()]
    for period in (2, 64, 1024, 16384):
        m = np.broadcast_to(top, (1, 1, len(moduli), n)).copy()
        m[..., (np.arange(n) // (period // 2)) % 2 == 1] = 0
        pats.append(m)
    spike = np.zeros((1, 1, len(moduli), n), dtype=np.uint64)
    spike[..., n - 1] = top[..., 0]
    pats.append(spike)
    x = np.concatenate(pats, axis=1)
    want = oracle.matrix_ntt(x, moduli)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), want)
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
    # the same patterns as evaluation-domain inputs of the inverse
    e = gpu.GpuDCRTPolyMatrix.from_rns(p, x, True)
    e.intt_all_in_place()
    assert np.array_equal(e.to_rns(), oracle.matrix_ntt(x, moduli, inverse=True))


def test_ntt_n16384_u64(gpu, oracle):
    n = 16384
    p = make_params(gpu, oracle, n, 2, 51, 17)
    moduli = p.moduli()
    x = rand_matrix(oracle, 5, 1, 2, moduli, n)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), oracle.matrix_ntt(x, moduli))
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)


@pytest.mark.parametrize("n,depth,bits,base", PARAM_SETS)
def test_add_sub_mul_scalar(gpu, oracle, n, depth, bits, base):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    a = rand_matrix(oracle, 10, 2, 3, moduli, n)
    b = rand_matrix(oracle, 11, 2, 3, moduli, n)
    s = rand_matrix(oracle, 12, 1, 1, moduli, n)
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    gs = gpu.GpuDCRTPolyMatrix.from_rns(p, s, True)
    assert np.array_equal((ga + gb).to_rns(), oracle.pointwise("add", a, b, moduli))
    assert np.array_equal((ga - gb).to_rns(), oracle.pointwise("sub", a, b, moduli))
    assert np.array_equal(ga.mul_scalar(gs).to_rns(), oracle.pointwise("mul", a, s, moduli))
    # ring axioms of the reference tests (gpu.rs:1258-1323)
    assert (ga - gb) + gb == ga
    assert np.array_equal((-ga).to_rns(), oracle.pointwise("sub", np.zeros_like(a), a, moduli))
    assert ga == ga.clone()
    assert not (ga == gb)


MATMUL_SHAPES = [(1, 1, 1), (1, 3, 5), (2, 2, 2), (3, 5, 4), (4, 7, 9), (5, 3, 17), (1, 30, 12), (8, 8, 8)]


@pytest.mark.parametrize("shape", MATMUL_SHAPES)
@pytest.mark.parametrize("n,depth,bits,base", [(16, 3, 18, 6), (128, 2, 16, 4), (256, 3, 51, 17), (4, 2, 17, 1), (64, 2, 30, 10)])
def test_matmul_vs_oracle(gpu, oracle, shape, n, depth, bits, base):
    r, k, c = shape
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    a = rand_matrix(oracle, 20, r, k, moduli, n)
    b = rand_matrix(oracle, 21, k, c, moduli, n)
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


@pytest.mark.parametrize("path", ["reg", "lds"])
@pytest.mark.parametrize("shape", [(8, 8, 8), (9, 5, 17), (16, 16, 16), (20, 7, 33), (1, 6, 40), (33, 3, 9)])
def test_matmul_kernel_families(gpu, oracle, hip_env, path, shape):
    """register-tiled and LDS-tiled products agree with the oracle on ragged shapes (n >= 64)."""
    r, k, c = shape
    n = 128
    p = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p.moduli()
    a = rand_matrix(oracle, 25, r, k, moduli, n)
    b = rand_matrix(oracle, 26, k, c, moduli, n)
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    hip_env.set("MXX_HIP_MATMUL_PATH", path)
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


@pytest.mark.parametrize("shape", [(32, 8, 16), (33, 12, 17), (64, 64, 64), (16, 4, 8), (40, 36, 5), (3, 16, 70), (32, 4, 32), (65, 20, 33)])
@pytest.mark.parametrize("bits", [24, 31])
@pytest.mark.parametrize("path", ["dma", "wide"])
def test_matmul_dma_kernel(gpu, oracle, hip_env, shape, bits, path):
    """global->LDS streamed products (matmul_dma.hip; "dma": 64 slots x 32x16 tile, "wide": 32 slots x 32x32 tile, 16 waves):
    full and ragged tiles, inner % 4 == 0, and 31-bit primes whose accumulators must be folded every chunk;
    worst-case residues q-1 in one operand."""
    r, k, c = shape
    n = 128
    moduli = oracle.gen_crt_basis(n, 2, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 8)
    a = rand_matrix(oracle, 27, r, k, moduli, n)
    b = np.broadcast_to((np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1), (k, c, len(moduli), n)).copy()
    b[::2] = rand_matrix(oracle, 28, k, c, moduli, n)[::2]
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    hip_env.set("MXX_HIP_MATMUL_PATH", path)
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


@pytest.mark.parametrize("tile", ["184", "184p", "144", "144p", "284", "284p", "244", "244p", "344", "344p", "444", "444p", "382", "382p", "481", "481p", "881", "881p", "482", "482p"])
@pytest.mark.parametrize("bits", [24, 31])
def test_matmul_register_tiles(gpu, oracle, hip_env, tile, bits):
    """every register-tile shape of matmul_kernel (rows x cols x slots per lane; p = the loads-ahead form with two
    operand sets) on ragged shapes: odd and even inner dimensions (the loads-ahead loop is unrolled by two), tiles that
    overhang both edges, worst-case residues, 31-bit primes (accumulators folded inside the loop)."""
    n = 64
    moduli = oracle.gen_crt_basis(n, 2, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 8)
    hip_env.set("MXX_HIP_MATMUL_PATH", "reg")
    hip_env.set("MXX_HIP_MATMUL_TILE", tile)
    for (r, k, c) in ((int(tile[0]), 1, int(tile[1])), (int(tile[0]) + 1, 6, 2 * int(tile[1]) + 1), (5, 37, 9), (1, 2, 1)):
        a = rand_matrix(oracle, 40, r, k, moduli, n)
        b = np.broadcast_to((np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1), (k, c, len(moduli), n)).copy()
        b[::2] = rand_matrix(oracle, 41, k, c, moduli, n)[::2]
        ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
        gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
        assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli)), (tile, r, k, c)


@pytest.mark.parametrize("shape", [(3, 5, 9), (3, 8, 4), (4, 7, 8), (4, 64, 17), (2, 33, 16), (1, 31, 15)])
def test_matmul_small_grid_dispatch(gpu, oracle, shape):
    """the shapes the automatic choice sends to the loads-ahead tiles (3 rows: 3x4x4; 4 rows and 1-2 rows on small grids)"""
    r, k, c = shape
    n = 256
    p = make_params(gpu, oracle, n, 3, 24, 12)
    moduli = p.moduli()
    a = rand_matrix(oracle, 42, r, k, moduli, n)
    b = rand_matrix(oracle, 43, k, c, moduli, n)
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


def test_matmul_lds_lazy_window_31bit(gpu, oracle):
    """LDS kernel with 31-bit primes: the 64-bit accumulators must be folded every chunk."""
    n = 64
    moduli = oracle.gen_crt_basis(n, 2, 31)
    p = gpu.GpuDCRTPolyParams(n, moduli, 8)
    top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1)
    a = np.broadcast_to(top, (9, 37, len(moduli), n)).copy()
    b = np.broadcast_to(top, (37, 10, len(moduli), n)).copy()
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


def test_matmul_worst_case_accumulation(gpu, oracle):
    """all residues = q-1, inner dimension past the lazy-accumulation window of 31-bit primes."""
    n = 16
    moduli = oracle.gen_crt_basis(n, 2, 31)
    p = gpu.GpuDCRTPolyParams(n, moduli, 8)
    k = 37
    top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1)
    a = np.broadcast_to(top, (2, k, len(moduli), n)).copy()
    b = np.broadcast_to(top, (k, 3, len(moduli), n)).copy()
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


def test_matmul_is_ring_product(gpu, oracle):
    """1x1 * 1x1 in EVAL == schoolbook negacyclic product of the coefficient forms."""
    n = 128
    p = make_params(gpu, oracle, n, 2, 17, 1)
    moduli = p.moduli()
    a = rand_matrix(oracle, 30, 1, 1, moduli, n)
    b = rand_matrix(oracle, 31, 1, 1, moduli, n)
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, False)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, False)
    ga.ntt_all_in_place()
    gb.ntt_all_in_place()
    c = (ga * gb).to_coeff_rns()
    for l, q in enumerate(moduli):
        assert np.array_equal(c[0, 0, l], oracle.negacyclic_schoolbook(a[0, 0, l], b[0, 0, l], q))


def test_golden_fixtures_on_gpu(gpu, oracle):
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    for f in sorted(os.listdir(gdir)):
        if not f.endswith(".npz") or f.startswith("samplers_"):  # the sampler fixtures: tests/test_gpu_sampling.py
            continue
        z = np.load(os.path.join(gdir, f))
        moduli = [int(q) for q in z["moduli"]]
        n, base = int(z["n"]), int(z["base_bits"])
        p = gpu.GpuDCRTPolyParams(n, moduli, base)
        a = gpu.GpuDCRTPolyMatrix.from_rns(p, z["a_coeff"], False)
        a.ntt_all_in_place()
        assert np.array_equal(a.to_rns(), z["a_eval"]), f
        b = gpu.GpuDCRTPolyMatrix.from_rns(p, z["b_eval"], True)
        assert np.array_equal((a * b).to_rns(), z["ab_eval"]), f
        m = gpu.GpuDCRTPolyMatrix.from_rns(p, z["m_coeff"], False)
        assert np.array_equal(m.decompose().to_coeff_rns(), z["m_decomposed"]), f
        assert np.array_equal(gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2).to_rns(), z["gadget_eval"]), f


@pytest.mark.parametrize("n,depth,bits,base", [(16, 2, 17, 1), (128, 2, 16, 4), (128, 2, 16, 8), (16, 3, 17, 5), (64, 2, 51, 17), (1024, 3, 24, 12)])
def test_decompose_and_gadget(gpu, oracle, n, depth, bits, base):
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    M = rand_matrix(oracle, 40, 2, 3, moduli, n)
    gm = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False)
    gm_eval = gm.ensure_eval()
    dec = gm_eval.decompose()  # EVAL in -> EVAL out
    want = oracle.decompose(M, moduli, base)
    assert dec.is_ntt
    assert np.array_equal(dec.to_coeff_rns(), want)
    assert np.array_equal(gm.decompose().to_coeff_rns(), want)  # COEFF in
    G = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2)
    assert G.size() == (2, 2 * p.modulus_digits())
    assert np.array_equal(G.to_rns(), oracle.gadget_matrix(2, moduli, n, base))
    # G * G^-1(M) == M   (gpu_dcrt_poly.rs:2075-2163)
    assert G * dec == gm_eval
    # small variants
    Gs = gpu.GpuDCRTPolyMatrix.small_gadget_matrix(p, 2)
    assert np.array_equal(Gs.to_rns(), oracle.gadget_matrix(2, moduli, n, base, small=True))
    sd = gm.small_decompose()
    assert np.array_equal(sd.to_coeff_rns(), oracle.decompose(M, moduli, base, small=True))


@pytest.mark.parametrize("depth,bits,base", [(3, 24, 12), (2, 24, 7), (2, 20, 20)])
def test_decompose_fused_with_ntt_at_2_14(gpu, oracle, hip_env, depth, bits, base):
    """n = 2^14 takes the fused digit + forward-NTT kernel (ntt14.h); it must give the bits of the
    two-step path and of the CPU restatement (digit layout, last-digit mask, small variant)."""
    n = 16384
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    M = rand_matrix(oracle, 41, 2, 2, moduli, n)
    gm = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False)
    want = oracle.matrix_ntt(oracle.decompose(M, moduli, base), moduli)
    fused = gm.decompose()
    assert fused.is_ntt and np.array_equal(fused.to_rns(), want)
    assert gm.ensure_eval().decompose() == fused  # EVAL source: private INTT copy feeds the fused kernel
    small = gm.small_decompose()
    assert np.array_equal(small.to_rns(), oracle.matrix_ntt(oracle.decompose(M, moduli, base, small=True), moduli))
    hip_env.set("MXX_HIP_DECOMPOSE_FUSED", "0")
    assert gm.decompose() == fused and gm.small_decompose() == small
    G = gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2)
    assert G * fused == gm.ensure_eval()


@pytest.mark.parametrize("logn,depth,bits,base", [
    (14, 2, 28, 14), (16, 2, 24, 12), (16, 3, 28, 14), (17, 2, 28, 9), (16, 2, 24, 24), (17, 2, 24, 12),
    (10, 3, 24, 12), (11, 2, 28, 14), (12, 2, 24, 7), (12, 2, 24, 24), (13, 2, 28, 14), (15, 2, 24, 12), (15, 2, 28, 9),
    (10, 3, 51, 17), (11, 2, 51, 51), (12, 2, 51, 17), (13, 2, 51, 20), (14, 2, 51, 17), (15, 2, 51, 17), (16, 2, 51, 25)])
def test_decompose_fused_with_ntt_tight_and_split_sizes(gpu, oracle, hip_env, logn, depth, bits, base):
    """The digit transform fused into the forward NTT's load beyond the 24-bit 2^14 case: 28-bit limbs at 2^14 (tight
    form of the grouped kernel); 2^16 / 2^17 points and 64-bit words from 2^15 (digits in the load of the head kernel,
    then the sub-vectors); every ring whose vector fits LDS (2^10..2^13, 2^15; 64-bit words 2^10..2^14) through
    ntt_fwd_lazy_digits_kernel; 24-, 28- and 51-bit limbs, a base as wide as a limb: against the CPU restatement and
    the two-step path, small variant, EVAL and COEFF sources, G * G^-1(M) = M."""
    n = 1 << logn
    moduli = oracle.gen_crt_basis(n, depth, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, base)
    M = rand_matrix(oracle, 141 + logn, 1, 2, moduli, n)
    M[0, 0, :, :3] = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(-1, 1)
    gm = gpu.GpuDCRTPolyMatrix.from_rns(p, M, False)
    want = oracle.matrix_ntt(oracle.decompose(M, moduli, base), moduli)
    fused = gm.decompose()
    assert fused.is_ntt and np.array_equal(fused.to_rns(), want)
    assert gm.ensure_eval().decompose() == fused
    small = gm.small_decompose()
    assert np.array_equal(small.to_rns(), oracle.matrix_ntt(oracle.decompose(M, moduli, base, small=True), moduli))
    hip_env.set("MXX_HIP_DECOMPOSE_FUSED", "0")
    assert gm.decompose() == fused and gm.small_decompose() == small
    hip_env.unset("MXX_HIP_DECOMPOSE_FUSED")
    assert gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 1) * fused == gm.ensure_eval()


def test_small_decomposed_identity_chunk(gpu, oracle):
    """chunked == full (gpu_dcrt_poly.rs:2225-2334)."""
    n, base = 16, 4
    p = make_params(gpu, oracle, n, 2, 16, base)
    k = -(-p.crt_bits() // base)
    size = 3
    scalar = gpu.GpuDCRTPoly.from_biguints(p, [5, 1, 7, 300])
    full = gpu.GpuDCRTPolyMatrix.identity(p, size, scalar).small_decompose()
    digits = scalar.inner.small_decompose()  # k x 1
    scalar_by_digit = [digits.entry(d, 0) for d in range(k)]
    for chunk in range(k):
        got = gpu.GpuDCRTPolyMatrix.small_decomposed_identity_chunk(p, size, chunk, k, scalar_by_digit)
        want = full.slice(chunk * size, (chunk + 1) * size, 0, size)
        assert got == want


def test_structure_ops(gpu, oracle):
    n = 16
    p = make_params(gpu, oracle, n, 3, 18, 6)
    moduli = p.moduli()
    a = rand_matrix(oracle, 50, 3, 4, moduli, n)
    b = rand_matrix(oracle, 51, 3, 2, moduli, n)
    c = rand_matrix(oracle, 52, 2, 4, moduli, n)
    ga, gb, gc = (gpu.GpuDCRTPolyMatrix.from_rns(p, x, True) for x in (a, b, c))
    assert np.array_equal(ga.slice(1, 3, 1, 4).to_rns(), a[1:3, 1:4])
    assert np.array_equal(ga.transpose().to_rns(), a.transpose(1, 0, 2, 3))
    assert np.array_equal(ga.concat_columns([gb]).to_rns(), np.concatenate([a, b], axis=1))
    assert np.array_equal(ga.concat_rows([gc]).to_rns(), np.concatenate([a, c], axis=0))
    diag = ga.concat_diag([gb]).to_rns()
    assert np.array_equal(diag[:3, :4], a) and np.array_equal(diag[3:, 4:], b)
    assert not diag[:3, 4:].any() and not diag[3:, :4].any()
    assert np.array_equal(ga.vectorize_columns().to_rns()[:, 0], a.transpose(1, 0, 2, 3).reshape(12, 3, n))
    # add_block / copy_block
    out = ga.clone()
    out.add_block_from(gc, 1, 0, 0, 0, 2, 4)
    want = a.copy()
    want[1:3] = oracle.pointwise("add", a[1:3], c, moduli)
    assert np.array_equal(out.to_rns(), want)
    out.copy_block_from(gb, 0, 2, 1, 0, 2, 2)
    want[0:2, 2:4] = b[1:3]
    assert np.array_equal(out.to_rns(), want)
    # identity and tensor
    I = gpu.GpuDCRTPolyMatrix.identity(p, 3)
    assert I * ga == ga
    t = I.tensor(gb).to_rns()
    for i in range(3):
        assert np.array_equal(t[3 * i : 3 * i + 3, 2 * i : 2 * i + 2], b)
    z = gpu.GpuDCRTPolyMatrix.zero(p, 2, 2)
    assert not z.to_rns().any()
    # entry / set_entry / const coeff
    e = ga.entry(2, 1)
    assert np.array_equal(e.inner.to_rns()[0, 0], a[2, 1])
    out2 = ga.clone()
    out2.set_entry(0, 0, e)
    assert np.array_equal(out2.to_rns()[0, 0], a[2, 1])
    coeff = ga.ensure_coeff()
    assert np.array_equal(coeff.store_const_coeff_words(), coeff.to_rns()[..., 0])


def test_mul_decompose_and_tensor_identity(gpu, oracle):
    n, base = 16, 6
    p = make_params(gpu, oracle, n, 3, 18, base)
    moduli = p.moduli()
    k = p.modulus_digits()
    S = rand_matrix(oracle, 60, 2, 3 * k, moduli, n)
    B = rand_matrix(oracle, 61, 3, 4, moduli, n)
    gs = gpu.GpuDCRTPolyMatrix.from_rns(p, S, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, B, True)
    want = oracle.matmul(S, oracle.matrix_ntt(oracle.decompose(oracle.matrix_ntt(B, moduli, inverse=True), moduli, base), moduli), moduli)
    assert np.array_equal(gs.mul_decompose(gb).to_rns(), want)
    os.environ["MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"] = "3"
    try:
        assert np.array_equal(gs.mul_decompose(gb).to_rns(), want)
    finally:
        del os.environ["MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"]
    # mul_tensor_identity: S' * (I_2 (x) B')
    S2 = rand_matrix(oracle, 62, 2, 6, moduli, n)
    gs2 = gpu.GpuDCRTPolyMatrix.from_rns(p, S2, True)
    want2 = np.concatenate([oracle.matmul(S2[:, 0:3], B, moduli), oracle.matmul(S2[:, 3:6], B, moduli)], axis=1)
    assert np.array_equal(gs2.mul_tensor_identity(gb, 2).to_rns(), want2)


def test_poly_wrappers(gpu, oracle):
    """from_coeffs -> coeffs round trip and cross-domain equality (gpu.rs:1239-1337)."""
    n = 16
    p = make_params(gpu, oracle, n, 3, 18, 6)
    Q = p.modulus()
    rng = np.random.default_rng(3)
    coeffs = [int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**62)) % Q for _ in range(n)]
    poly = gpu.GpuDCRTPoly.from_biguints(p, coeffs)
    assert poly.is_ntt()
    assert poly.coeffs() == coeffs
    assert poly.ensure_coeff_domain() == poly
    one = gpu.GpuDCRTPoly.const_one(p)
    assert (poly * one) == poly
    assert (poly + gpu.GpuDCRTPoly.const_zero(p)) == poly
    assert ((poly - poly).coeffs()) == [0] * n
    m1 = gpu.GpuDCRTPoly.const_minus_one(p)
    assert (poly * m1) == -poly


def test_error_behaviour(gpu, oracle):
    """Non-zero status -> exception with the reference's message shape (gpu.rs:249-263)."""
    n = 16
    p = make_params(gpu, oracle, n, 3, 18, 6)
    moduli = p.moduli()
    a = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 70, 2, 2, moduli, n), False)
    b = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 71, 2, 2, moduli, n), True)
    from mxx_amd import _ffi
    import ctypes as C

    out = gpu.GpuDCRTPolyMatrix.new_empty(p, 2, 2)
    assert _ffi.lib().gpu_matrix_mul(out.raw, a.raw, b.raw) != 0
    assert "requires Eval format" in _ffi.last_error_string()
    assert _ffi.lib().gpu_matrix_mul_scalar(out.raw, a.raw, b.raw) != 0
    buf = np.zeros((2, 2, 3, n), dtype=np.uint64)
    ev = C.c_void_p()
    st = _ffi.lib().gpu_matrix_store_rns_batch(a.raw, buf.ctypes.data, 3 * n * 8, _ffi.GPU_POLY_FORMAT_EVAL, C.byref(ev))
    assert st != 0 and "format conversion is not supported" in _ffi.last_error_string()
    st = _ffi.lib().gpu_matrix_store_const_coeff_batch(b.raw, buf.ctypes.data, 3, C.byref(ev))
    assert st != 0
    # equal: mismatching format is "not equal", not an error
    eq = C.c_int(7)
    assert _ffi.lib().gpu_matrix_equal(a.raw, b.raw, C.byref(eq)) == 0 and eq.value == 0
    with pytest.raises(AssertionError):
        gpu.GpuDCRTPolyMatrix(p, 1, 1, 5, True)  # invalid level (gpu_dcrt_poly.rs:229)
    raw = C.c_void_p()
    assert _ffi.lib().gpu_matrix_create(p.ctx_raw(), 5, 1, 1, 1, C.byref(raw)) != 0


def test_empty_and_degenerate_shapes(gpu, oracle):
    """0 x c, r x 0 and 1 x 1 operands through every family of entry points: no launch, no error, right
    shapes and format tags (the reference treats zero-sized matrices as ordinary values)."""
    n = 64
    p = make_params(gpu, oracle, n, 2, 24, 12)
    M = gpu.GpuDCRTPolyMatrix
    k = p.modulus_digits()
    for r, c in ((0, 3), (3, 0), (0, 0)):
        z = M.zero(p, r, c)
        assert z.size() == (r, c)
        assert (z + z) == z and (z - z) == z
        z2 = z.clone()
        z2.ntt_all_in_place()
        z2.intt_all_in_place()
        assert z2.size() == (r, c)
        assert z.to_rns().shape == (r, c, 2, n)
        d = z.decompose()
        assert d.size() == (r * k, c) and d.is_ntt
        assert z.small_decompose().size()[1] == c
        s = M.sample_distribution(p, r, c, oracle.DIST["gauss"], 3.0, gpu.GpuRngSeed.from_bytes(bytes(32)))
        assert s.size() == (r, c) and s.is_ntt
        assert z.transpose().size() == (c, r)
        payload = z.to_compact_bytes()
        assert M.from_compact_bytes(p, payload).size() == (r, c)
    a = M.from_rns(p, rand_matrix(oracle, 90, 2, 3, p.moduli(), n), True)
    assert (M.zero(p, 0, 2).ensure_eval() * a).size() == (0, 3)            # 0 x 2 times 2 x 3
    assert (a * M.zero(p, 3, 0).ensure_eval()).size() == (2, 0)            # 2 x 3 times 3 x 0
    inner0 = M.zero(p, 2, 0).ensure_eval() * M.zero(p, 0, 4).ensure_eval()  # empty sum: the zero matrix
    assert inner0 == M.zero(p, 2, 4).ensure_eval()
    assert a.slice(1, 1, 0, 3).size() == (0, 3) and a.slice(0, 2, 2, 2).size() == (2, 0)
    assert a.concat_columns([M.zero(p, 2, 0).ensure_eval()]) == a
    one = M.from_rns(p, rand_matrix(oracle, 91, 1, 1, p.moduli(), n), True)
    assert np.array_equal((one * one).to_rns(), oracle.matmul(one.to_rns(), one.to_rns(), p.moduli()))
    # zero-column preimage: the sampler returns an empty (k+2) x 0 matrix
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, A = sampler.trapdoor(p, 1)
    x = sampler.preimage(p, td, A, M.zero(p, 1, 0).ensure_eval())
    assert x.size() == (k + 2, 0)


def test_maximum_limb_count(gpu, oracle):
    """64 limbs (GPUPOLY_MAX_LIMBS, Runtime.cuh:51 of the reference): NTT, product, decompose, gadget
    relation and the compact wire format (1280-bit coefficients through the on-device Garner CRT)."""
    n, depth, bits, base = 16, 64, 20, 10
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    assert len(moduli) == 64
    M = gpu.GpuDCRTPolyMatrix
    a = rand_matrix(oracle, 95, 2, 3, moduli, n)
    b = rand_matrix(oracle, 96, 3, 2, moduli, n)
    ga, gb = M.from_rns(p, a, False), M.from_rns(p, b, False)
    ga.ntt_all_in_place()
    assert np.array_equal(ga.to_rns(), oracle.matrix_ntt(a, moduli))
    gb.ntt_all_in_place()
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(oracle.matrix_ntt(a, moduli), oracle.matrix_ntt(b, moduli), moduli))
    dec = M.from_rns(p, a, False).decompose()
    assert np.array_equal(dec.to_coeff_rns(), oracle.decompose(a, moduli, base))
    assert M.gadget_matrix(p, 2) * dec == ga
    back = M.from_compact_bytes(p, ga.to_compact_bytes())
    assert back == ga
    s = M.sample_distribution(p, 1, 2, oracle.DIST["uniform"], 0.0, gpu.GpuRngSeed.from_bytes(bytes(range(32))))
    assert np.array_equal(s.to_coeff_rns(), oracle.sample_distribution(1, 2, moduli, n, "uniform", 0.0, bytes(range(32))))


@pytest.mark.parametrize("logn", [15, 16, 17])
def test_ntt_large_ring_dimensions(gpu, oracle, logn):
    """n = 2^15 (largest LDS-resident kernel), 2^16 and 2^17 (head / tail kernel + LDS sub-transforms; 2^17 is the limit)."""
    n = 1 << logn
    moduli = oracle.gen_crt_basis(n, 2, 24)
    p = gpu.GpuDCRTPolyParams(n, moduli, 12)
    x = rand_matrix(oracle, 97, 1, 2, moduli, n)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), oracle.matrix_ntt(x, moduli))
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
    y = rand_matrix(oracle, 98, 2, 1, moduli, n)
    gy = gpu.GpuDCRTPolyMatrix.from_rns(p, y, False).ensure_eval()
    assert np.array_equal((m.ensure_eval() * gy).to_rns(), oracle.matmul(oracle.matrix_ntt(x, moduli), oracle.matrix_ntt(y, moduli), moduli))
    # the callers of the transform at this size: decompose (EVAL in/out), Gaussian matrix, compact bytes
    dec = gy.decompose()
    assert np.array_equal(dec.to_coeff_rns(), oracle.decompose(y, moduli, 12))
    assert gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2) * dec == gy
    sd = bytes(range(32))
    g = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 1, 1, oracle.DIST["gauss"], 7.5, gpu.GpuRngSeed.from_bytes(sd))
    assert np.array_equal(g.to_coeff_rns(), oracle.sample_distribution(1, 1, moduli, n, "gauss", 7.5, sd))
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, gy.to_compact_bytes()) == gy


@pytest.mark.parametrize("bits", [26, 27, 28])
@pytest.mark.parametrize("logn", [10, 11, 12, 13, 14, 15, 16, 17])
def test_ntt_28_bit_moduli_tight_lazy_kernels(gpu, oracle, hip_env, logn, bits):
    """26..28-bit limbs in 32-bit words (the reference's end-to-end parameter sets: crt_bits = 28 at n = 2^16,
    tests/test_gpu_diamond_io.rs:64-70): the lazy kernels' tight forms (16 q <= 2^32: forward passes re-centre their
    inputs, inverse passes cap the bound exponents at 4) against the oracle and the fully reduced kernels, on the
    largest such primes, with extreme inputs (all q - 1, alternating 0 / q - 1 at several periods, a spike)."""
    n = 1 << logn
    moduli = oracle.gen_crt_basis(n, 2, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 14)
    top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1)
    pats = [rand_matrix(oracle, 120 + logn, 1, 1, moduli, n), np.broadcast_to(top, (1, 1, len(moduli), n)).copy()]
    for period in (2, 32, 1024, n):
        m = np.broadcast_to(top, (1, 1, len(moduli), n)).copy()
        m[..., (np.arange(n) // (period // 2)) % 2 == 1] = 0
        pats.append(m)
    spike = np.zeros((1, 1, len(moduli), n), dtype=np.uint64)
    spike[..., n - 1] = top[..., 0]
    pats.append(spike)
    x = np.concatenate(pats, axis=1)
    want = oracle.matrix_ntt(x, moduli)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), want)
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
    e = gpu.GpuDCRTPolyMatrix.from_rns(p, x, True)  # the same patterns as evaluation-domain inputs of the inverse
    e.intt_all_in_place()
    assert np.array_equal(e.to_rns(), oracle.matrix_ntt(x, moduli, inverse=True))
    hip_env.set("MXX_HIP_NTT_PATH", "global" if logn > 15 else "generic")
    g = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    g.ntt_all_in_place()
    assert np.array_equal(g.to_rns(), want)
    hip_env.unset("MXX_HIP_NTT_PATH")
    if logn == 14:
        hip_env.set("MXX_HIP_NTT14", "whole")
        w = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
        w.ntt_all_in_place()
        assert np.array_equal(w.to_rns(), want)
        w.intt_all_in_place()
        assert np.array_equal(w.to_rns(), x)


def test_28_bit_ring_whole_path(gpu, oracle):
    """n = 2^16, 28-bit limbs: product, decompose, Gaussian matrix and compact bytes ride on the tight transforms."""
    n = 1 << 16
    moduli = oracle.gen_crt_basis(n, 3, 28)
    p = gpu.GpuDCRTPolyParams(n, moduli, 14)
    x = rand_matrix(oracle, 131, 1, 2, moduli, n)
    y = rand_matrix(oracle, 132, 2, 1, moduli, n)
    gx = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False).ensure_eval()
    gy = gpu.GpuDCRTPolyMatrix.from_rns(p, y, False).ensure_eval()
    prod = gx * gy
    assert np.array_equal(prod.to_rns(), oracle.matmul(oracle.matrix_ntt(x, moduli), oracle.matrix_ntt(y, moduli), moduli))
    dec = gy.decompose()
    assert np.array_equal(dec.to_coeff_rns(), oracle.decompose(y, moduli, 14))
    assert gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 2) * dec == gy
    sd = bytes(range(32))
    g = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 1, 1, oracle.DIST["gauss"], 7.5, gpu.GpuRngSeed.from_bytes(sd))
    assert np.array_equal(g.to_coeff_rns(), oracle.sample_distribution(1, 1, moduli, n, "gauss", 7.5, sd))
    assert gpu.GpuDCRTPolyMatrix.from_compact_bytes(p, prod.to_compact_bytes()) == prod
    # the reference's trapdoor predicates on this ring (src/sampler/trapdoor/gpu.rs:558-633)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    trapdoor, public_matrix = sampler.trapdoor(p, 1)
    target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, 2, gpu.DistType.FinRingDist())
    preimage = sampler.preimage(p, trapdoor, public_matrix, target)
    assert public_matrix * preimage == target
    assert gx.mul_scalar_intt(gpu.GpuDCRTPolyMatrix.from_rns(p, y[:1], False).ensure_eval()) == (gx.mul_scalar(gpu.GpuDCRTPolyMatrix.from_rns(p, y[:1], False).ensure_eval())).into_coeff_domain()


@pytest.mark.parametrize("logn,bits", [(15, 51), (16, 51), (17, 51), (16, 24), (17, 24)])
def test_ntt_beyond_lds_split_kernels_equal_the_per_stage_path(gpu, oracle, hip_env, logn, bits):
    """Rings whose vectors do not fit LDS (u32 from 2^16 points, u64 from 2^15): the two-kernel transform (outer stages
    on strided sets + LDS kernel on the sub-vectors) against the oracle and against the one-launch-per-stage kernels
    (MXX_HIP_NTT_PATH=global), both directions, extreme inputs included."""
    n = 1 << logn
    moduli = oracle.gen_crt_basis(n, 2, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 12)
    x = rand_matrix(oracle, 99, 2, 1, moduli, n)
    x[0, 0, :, :4] = 0
    x[1, 0, :, -4:] = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(-1, 1)
    want = oracle.matrix_ntt(x, moduli)
    m = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    assert np.array_equal(m.to_rns(), want)
    hip_env.set("MXX_HIP_NTT_PATH", "global")
    g = gpu.GpuDCRTPolyMatrix.from_rns(p, x, False)
    g.ntt_all_in_place()
    assert g == m
    g.intt_all_in_place()
    assert np.array_equal(g.to_rns(), x)
    hip_env.unset("MXX_HIP_NTT_PATH")
    m.intt_all_in_place()
    assert np.array_equal(m.to_rns(), x)
