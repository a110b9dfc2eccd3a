"""GPU: ABI entries and wrapper methods that are bound but were not exercised anywhere else, BASELINE configs[0]
at its own size, and the Rust wrapper's add -> retag -> ntt sequence replayed through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu


def _seed(gpu, i):
    return gpu.GpuRngSeed.from_bytes(bytes([(i * 29 + j) & 0xFF for j in range(32)]))


@pytest.mark.parametrize("d,n,bits,cols", [(1, 64, 24, 3), (2, 32, 51, 2), (5, 16, 24, 2)])
def test_sample_p1_full_equals_cached(gpu, oracle, d, n, bits, cols):
    """gpu_matrix_sample_p1_full (cuda/include/matrix/MatrixTrapdoor.cuh:66-101; declared, not bound by the Rust
    side): factorisation + sampling in one call == covariance cache + sample_p1_full_cached, same seed."""
    from mxx_amd import _ffi

    base = 12 if bits == 24 else 17
    p = make_params(gpu, oracle, n, 2, bits, base)
    moduli = p.moduli()
    rng = np.random.default_rng(d * 7 + n)

    def small(rows, c, bound):
        v = rng.integers(-bound, bound + 1, size=(rows, c, n))
        return np.stack([np.mod(v, q).astype(np.uint64) for q in moduli], axis=-2)

    r, e = small(d, 2 * d, 6), small(d, 2 * d, 6)
    gr, ge = gpu.GpuDCRTPolyMatrix.from_rns(p, r, False), gpu.GpuDCRTPolyMatrix.from_rns(p, e, False)
    gr.ntt_all_in_place()
    ge.ntt_all_in_place()
    a = (gr * gr.transpose()).into_coeff_domain()
    b = (gr * ge.transpose()).into_coeff_domain()
    dm = (ge * ge.transpose()).into_coeff_domain()
    tp2 = gpu.GpuDCRTPolyMatrix.from_rns(p, small(2 * d, cols, 2000), False)
    sigma, s_par, dgg = 4.578 * 4097, 9.0e6, 4.578
    seed = _seed(gpu, 3)
    cache = gpu.GpuDCRTPolyMatrix.create_p1_covariance_cache(a, b, dm, sigma, s_par, dgg)
    want = gpu.GpuDCRTPolyMatrix.sample_p1_full_cached(cache, tp2.clone(), seed)
    out = gpu.GpuDCRTPolyMatrix.new_empty(p, 2 * d, cols)
    st = _ffi.lib().gpu_matrix_sample_p1_full(a.raw, b.raw, dm.raw, tp2.raw, sigma, s_par, dgg, seed, out.raw)
    _ffi.check_status(st, "gpu_matrix_sample_p1_full")
    out.is_ntt = True
    assert out == want


def test_poly_compact_bytes_aliases(gpu, oracle):
    """gpu_poly_{store,load}_compact_bytes: the 1x1 aliases of the matrix entry points (MatrixSerde.cuh:10-58)."""
    from mxx_amd import _ffi

    n = 128
    p = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p.moduli()
    coeff = rand_matrix(oracle, 301, 1, 1, moduli, n)
    lib = _ffi.lib()

    def store(fn, m):
        cap = n * 16 + 16
        buf = (C.c_uint8 * cap)()
        bits, bpc, ln = C.c_uint16(0), C.c_uint16(0), C.c_size_t(0)
        _ffi.check_status(fn(m.raw, buf, cap, C.byref(bits), C.byref(bpc), C.byref(ln)), "store_compact_bytes")
        return bytes(buf[: ln.value]), bits.value, bpc.value

    m1 = gpu.GpuDCRTPolyMatrix.from_rns(p, coeff, False)
    m2 = gpu.GpuDCRTPolyMatrix.from_rns(p, coeff, False)
    got = store(lib.gpu_poly_store_compact_bytes, m1)
    assert got == store(lib.gpu_matrix_store_compact_bytes, m2)
    assert got == oracle.compact_payload(coeff, moduli)
    back = gpu.GpuDCRTPolyMatrix.new_empty(p, 1, 1)
    payload = (C.c_uint8 * len(got[0])).from_buffer_copy(got[0])
    _ffi.check_status(lib.gpu_poly_load_compact_bytes(back.raw, payload, len(got[0]), got[1]), "gpu_poly_load_compact_bytes")
    back.is_ntt = False
    assert np.array_equal(back.to_rns(), coeff)


@pytest.mark.parametrize("n,depth,bits,base", [(128, 2, 24, 12), (64, 3, 17, 5), (32, 2, 51, 17)])
def test_mul_decompose_small_and_column_helpers(gpu, oracle, n, depth, bits, base):
    """mul_decompose_small, get_column_matrix_decompose, mul_tensor_identity_decompose
    (src/matrix/gpu_dcrt_poly.rs:1495-1579) against the CPU restatement's decompose + product."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    k = p.modulus_digits()
    ks = -(-bits // base)
    rows_b, cols_b = 2, 3
    # small-norm B (the small variants assume |B| < min q_i: limb 0 carries the value, matrix/mod.rs:251-255)
    rng = np.random.default_rng(n)
    v = rng.integers(0, 1 << (bits - 2), size=(rows_b, cols_b, n))
    b_small = np.stack([np.mod(v, q).astype(np.uint64) for q in moduli], axis=-2)
    s_small = oracle.matrix_ntt(rand_matrix(oracle, 7, 2, rows_b * ks, moduli, n), moduli)
    gs = gpu.GpuDCRTPolyMatrix.from_rns(p, s_small, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b_small, False)
    dec_small = oracle.matrix_ntt(oracle.decompose(b_small, moduli, base, small=True), moduli)
    assert np.array_equal(gs.mul_decompose_small(gb).to_rns(), oracle.matmul(s_small, dec_small, moduli))
    # full-gadget column helper
    b = rand_matrix(oracle, 8, rows_b, cols_b, moduli, n)
    gbf = gpu.GpuDCRTPolyMatrix.from_rns(p, b, False)
    dec = oracle.matrix_ntt(oracle.decompose(b, moduli, base), moduli)
    for j in range(cols_b):
        assert np.array_equal(gbf.get_column_matrix_decompose(j).to_rns(), dec[:, j : j + 1])
    # S * (I_m (x) G^-1(B)): block i of S (rows_b*k columns) times every decomposed column of B
    ident = 2
    s = oracle.matrix_ntt(rand_matrix(oracle, 9, 2, rows_b * k * ident, moduli, n), moduli)
    gS = gpu.GpuDCRTPolyMatrix.from_rns(p, s, True)
    got = gS.mul_tensor_identity_decompose(gbf.ensure_eval(), ident)
    w = rows_b * k
    want = np.concatenate([oracle.matmul(np.ascontiguousarray(s[:, i * w : (i + 1) * w]), dec, moduli) for i in range(ident)], axis=1)
    assert np.array_equal(got.to_rns(), want)


def test_add_then_ntt_as_the_rust_wrapper_sequences_it(gpu, oracle):
    """The Rust wrapper's `add` calls gpu_matrix_add and then copies the rhs's domain flag into its own
    (src/matrix/gpu_dcrt_poly.rs:406-408); a later ntt_all_in_place must really transform.  The reference's C side
    tags the sum EVAL whatever the operands were (MatrixArith.cu:2694), which would turn that NTT into a no-op on
    coefficient-domain data; here add/sub keep the operands' tag, so the sequence
    add(COEFF, COEFF) -> ntt_all gives NTT(a + b)."""
    from mxx_amd import _ffi

    n = 256
    p = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p.moduli()
    a, b = rand_matrix(oracle, 51, 2, 2, moduli, n), rand_matrix(oracle, 52, 2, 2, moduli, n)
    ga, gb = gpu.GpuDCRTPolyMatrix.from_rns(p, a, False), gpu.GpuDCRTPolyMatrix.from_rns(p, b, False)
    lib = _ffi.lib()
    out = gpu.GpuDCRTPolyMatrix.new_empty(p, 2, 2)          # created EVAL, like new_empty on the Rust side
    _ffi.check_status(lib.gpu_matrix_add(out.raw, ga.raw, gb.raw), "gpu_matrix_add")
    out.is_ntt = gb.is_ntt                                  # gpu_dcrt_poly.rs:406-408
    _ffi.check_status(lib.gpu_matrix_ntt_all(out.raw), "gpu_matrix_ntt_all")
    out.is_ntt = True
    want = oracle.matrix_ntt(oracle.pointwise("add", a, b, moduli), moduli)
    assert np.array_equal(out.to_rns(), want)
    # and the same through sub, then back
    _ffi.check_status(lib.gpu_matrix_sub(out.raw, ga.raw, gb.raw), "gpu_matrix_sub")
    out.is_ntt = gb.is_ntt
    assert np.array_equal(out.to_rns(), oracle.pointwise("sub", a, b, moduli))


def test_config0_plumbing_shape_against_the_oracle(gpu, oracle):
    """BASELINE.json configs[0]: n = 2^12, 2 RNS limbs, 4x4 * 4x4 (the CPU bench's plumbing case) - small enough
    for the CPU restatement: product, decompose and G * G^-1 on that ring, bit for bit."""
    n = 4096
    p = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p.moduli()
    a = oracle.matrix_ntt(rand_matrix(oracle, 401, 4, 4, moduli, n), moduli)
    b = oracle.matrix_ntt(rand_matrix(oracle, 402, 4, 4, moduli, n), moduli)
    ga, gb = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True), gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    prod = ga * gb
    want = oracle.matmul(a, b, moduli)
    assert np.array_equal(prod.to_rns(), want)
    coeff = oracle.matrix_ntt(want, moduli, inverse=True)
    assert np.array_equal(prod.to_coeff_rns(), coeff)
    dec = prod.decompose()
    assert np.array_equal(dec.to_rns(), oracle.matrix_ntt(oracle.decompose(coeff, moduli, 12), moduli))
    assert gpu.GpuDCRTPolyMatrix.gadget_matrix(p, 4) * dec == prod


def test_copy_to_context_and_sharded_preimages_two_contexts_one_device(gpu, oracle):
    """Device-to-device replica transport (gpupoly_matrix_copy_to_context) and the concurrent fan-out of
    preimage_batched_sharded (src/sampler/trapdoor/gpu.rs:371-397), rehearsed with two contexts on the one GPU
    this box has (the driver's 8-GPU node gives them different devices; `params_for_device` names them)."""
    n = 256
    p0 = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p0.moduli()
    # a second context on the same device: same ring, different dnum -> its own context / stream / cache
    p1 = gpu.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=p0.gpu_ids(), dnum=2)
    assert p1.ctx_raw().value != p0.ctx_raw().value
    assert p0.params_for_device(p0.gpu_ids()[0]).ctx().device() == p0.ctx().device()
    x = rand_matrix(oracle, 600, 3, 2, moduli, n)
    m0 = gpu.GpuDCRTPolyMatrix.from_rns(p0, x, False)
    m0.ntt_all_in_place()                      # still in flight on context 0's stream when the copy is enqueued
    m1 = m0.to_params(p1)
    assert m1.params is p1 and m1.is_ntt and np.array_equal(m1.to_rns(), oracle.matrix_ntt(x, moduli))
    del m0                                     # stream-ordered free of the source after the copy
    assert np.array_equal((m1 + m1).to_rns(), oracle.pointwise("add", oracle.matrix_ntt(x, moduli), oracle.matrix_ntt(x, moduli), moduli))
    # trapdoor replica + sharded preimages: requests alternate between the two contexts
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p0, 4.578)
    td0, a0 = sampler.trapdoor(p0, 1)
    td1, a1 = td0.to_params(p1), a0.to_params(p1)
    assert td1.r.params is p1 and np.array_equal(td1.a_mat_coeff.to_rns(), td0.a_mat_coeff.to_rns())
    us = gpu.GpuDCRTPolyUniformSampler()
    reqs = []
    for i in range(6):
        p, td, a = (p0, td0, a0) if i % 2 == 0 else (p1, td1, a1)
        reqs.append((100 + i, p, td, a, us.sample_uniform(p, 1, 2 + i % 3, gpu.DistType.FinRingDist())))
    outs = sampler.preimage_batched_sharded(reqs)
    assert [idx for idx, _ in outs] == [100 + i for i in range(6)]
    for (idx, xx), (_, p, td, a, t) in zip(outs, reqs):
        assert xx.params is p and a * xx == t


@pytest.mark.parametrize("n,depth,bits,design", [(16384, 3, 24, "grouped"), (16384, 2, 24, "unsigned"), (16384, 2, 25, "grouped"),
                                                 (16384, 2, 24, "whole"), (16384, 2, 28, "grouped"), (1024, 2, 24, "grouped"),
                                                 (2048, 2, 28, "grouped"), (4096, 3, 24, "grouped"), (8192, 2, 27, "grouped"),
                                                 (32768, 2, 24, "grouped"), (65536, 2, 28, "grouped"), (65536, 2, 24, "grouped"),
                                                 (131072, 2, 28, "grouped"), (512, 2, 24, "grouped"), (256, 2, 51, "grouped")])
def test_mul_scalar_intt_fused(gpu, oracle, hip_env, n, depth, bits, design):
    """gpupoly_matrix_mul_scalar_intt: INTT(x o w) with the product (a Montgomery one, on w's plain residues) in the
    inverse transform's load: the grouped 2^14 kernels (signed and unsigned butterflies), every whole-vector LDS kernel
    (2^10..2^15, lazy and 26..28-bit TIGHT forms), the split 2^16 / 2^17 kernels (the reference's end-to-end ring,
    tests/test_gpu_diamond_io.rs:64-71) - and the two-call fallback elsewhere (n < 2^10, 64-bit words).  Bit-exact
    against the CPU restatement, out-of-place and in place."""
    from mxx_amd import _ffi

    hip_env.set("MXX_HIP_NTT14", design)
    p = make_params(gpu, oracle, n, depth, bits, 12 if bits <= 25 else 17)
    moduli = p.moduli()
    x = oracle.matrix_ntt(rand_matrix(oracle, 700, 2, 3, moduli, n), moduli)
    w = oracle.matrix_ntt(rand_matrix(oracle, 701, 1, 1, moduli, n), moduli)
    x[0, 0] = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(-1, 1)  # worst-case residues q-1
    gx, gw = gpu.GpuDCRTPolyMatrix.from_rns(p, x, True), gpu.GpuDCRTPolyMatrix.from_rns(p, w, True)
    want = oracle.matrix_ntt(oracle.pointwise("mul", x, w, moduli), moduli, inverse=True)
    out = gx.mul_scalar_intt(gw)
    assert not out.is_ntt and np.array_equal(out.to_rns(), want)
    assert np.array_equal(gx.to_rns(), x)  # the operand is untouched
    _ffi.check_status(_ffi.lib().gpupoly_matrix_mul_scalar_intt(gx.raw, gx.raw, gw.raw), "gpupoly_matrix_mul_scalar_intt")
    gx.is_ntt = False
    assert np.array_equal(gx.to_rns(), want)


@pytest.mark.parametrize("shape", [(32, 32, 32), (64, 64, 64), (33, 12, 40), (40, 100, 70), (3, 16, 70), (64, 128, 32), (70, 5, 33)])
@pytest.mark.parametrize("bits", [24, 22, 17])
def test_matmul_mfma_kernel(gpu, oracle, hip_env, shape, bits):
    """Matrix-core form of the R_q product (matmul_mfma.hip: centred residues as three balanced int8 digits,
    v_mfma_i32_32x32x32_i8 into five significance planes, 64-bit recombination + Barrett): bit-exact against the
    CPU restatement on full and ragged 32x32 tiles, inner dimensions that are not multiples of 32, and the
    residues at the centring boundary (0, (q-1)/2, (q+1)/2, q-1)."""
    r, k, c = shape
    n = 128
    moduli = oracle.gen_crt_basis(n, 2, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 8)
    a = rand_matrix(oracle, 27, r, k, moduli, n)
    b = rand_matrix(oracle, 28, k, c, moduli, n)
    qs = np.asarray(moduli, dtype=np.uint64).reshape(-1, 1)
    edge = np.concatenate([np.zeros_like(qs), (qs - 1) // 2, (qs + 1) // 2, qs - 1], axis=1)  # (L, 4)
    a[0, :, :, :4] = edge
    b[:, 0, :, :4] = edge[:, ::-1]
    a[-1, :, :, 4:8] = (qs - 1)
    b[:, -1, :, 4:8] = (qs - 1)
    ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    hip_env.set("MXX_HIP_MATMUL_PATH", "mfma")
    assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


def test_matmul_mfma_equals_valu_at_full_size(gpu, oracle, hip_env):
    """64 x 64 x 64 at n = 2^14, L = 8 (BASELINE configs[2]): the matrix-core kernel and the VALU kernel agree."""
    p = make_params(gpu, oracle, 16384, 8, 24, 12)
    us = gpu.GpuDCRTPolyUniformSampler()
    a = us.sample_uniform(p, 64, 64, gpu.DistType.FinRingDist())
    b = us.sample_uniform(p, 64, 64, gpu.DistType.FinRingDist())
    hip_env.set("MXX_HIP_MATMUL_PATH", "dma")
    want = a * b
    hip_env.set("MXX_HIP_MATMUL_PATH", "mfma")
    assert a * b == want
    hip_env.set("MXX_HIP_MATMUL_PATH", "wide")
    assert a * b == want


@pytest.mark.parametrize("shape,n", [((2, 72, 4), 256), ((5, 9, 7), 256), ((1, 3, 1), 64), ((8, 40, 9), 4096), ((3, 70, 5), 16)])
def test_matmul_u64_lazy_128bit_accumulators(gpu, oracle, shape, n):
    """64-bit words (BASELINE configs[4]: 51-bit limbs): 128-bit lazy accumulators with one three-step Barrett
    reduction per output, every register-tile configuration the launcher picks by occupancy, worst-case residues
    q-1 in both operands, and a small prime riding in a wide context (the k < 33 branch of reduce_u128_sum)."""
    r, k, c = shape
    big = oracle.gen_crt_basis(n, 2, 51)
    small = next(q for q in range(2 * n + 1, 1 << 20, 2 * n) if oracle.lib().orc_is_prime(q))
    for moduli in (big, [big[0], small], oracle.gen_crt_basis(n, 1, 60)):
        p = gpu.GpuDCRTPolyParams(n, moduli, 17)
        assert p.ctx().word_bytes() == 8
        a = rand_matrix(oracle, 31, r, k, moduli, n)
        b = rand_matrix(oracle, 32, k, c, moduli, n)
        top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(-1, 1)
        a[0], b[:, 0] = top, top
        ga, gb = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True), gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
        assert np.array_equal((ga * gb).to_rns(), oracle.matmul(a, b, moduli))


@pytest.mark.parametrize("rows_s,rows_b,cols_b,eval_b", [(2, 2, 3, True), (8, 5, 4, False)])
def test_mul_decompose_one_call_at_2_14(gpu, oracle, rows_s, rows_b, cols_b, eval_b):
    """gpupoly_matrix_mul_decompose at n = 2^14 (digits generated inside the forward transform, all columns at once,
    product straight into the output; what GpuDCRTPolyMatrix.mul_decompose calls by default): bit-exact against the
    CPU restatement's decompose + product, for COEFF and EVAL operands."""
    n, base = 16384, 12
    p = make_params(gpu, oracle, n, 2, 24, base)
    moduli = p.moduli()
    k = p.modulus_digits()
    S = oracle.matrix_ntt(rand_matrix(oracle, 801, rows_s, rows_b * k, moduli, n), moduli)
    Bc = rand_matrix(oracle, 802, rows_b, cols_b, moduli, n)
    want = oracle.matmul(S, oracle.matrix_ntt(oracle.decompose(Bc, moduli, base), moduli), moduli)
    gs = gpu.GpuDCRTPolyMatrix.from_rns(p, S, True)
    gb = gpu.GpuDCRTPolyMatrix.from_rns(p, Bc, False)
    if eval_b:
        gb.ntt_all_in_place()
    keep = gb.to_rns().copy()
    got = gs.mul_decompose(gb)
    assert np.array_equal(got.to_rns(), want)
    assert np.array_equal(gb.to_rns(), keep) and gb.is_ntt == eval_b  # the operand is untouched


@pytest.mark.parametrize("path", ["reg", "lds", "dma"])
def test_matmul_lazy_window_worst_case_both_operands(gpu, oracle, hip_env, path):
    """31-bit primes, every residue of BOTH operands q-1, inner dimension far beyond the lazy window: the 64-bit
    accumulators carry a folded residue into every window of products (runtime.hip: lazy_terms = (2^64 - q) / (q-1)^2;
    the round-1 formula (2^64 - 1) / (q-1)^2 ignored the carried residue)."""
    n, r, k, c = 128, 32, 96, 16
    moduli = oracle.gen_crt_basis(n, 2, 31)
    p = gpu.GpuDCRTPolyParams(n, moduli, 8)
    top = (np.asarray(moduli, dtype=np.uint64) - np.uint64(1)).reshape(1, 1, -1, 1)
    a = np.broadcast_to(top, (r, k, len(moduli), n)).copy()
    b = np.broadcast_to(top, (k, c, len(moduli), n)).copy()
    hip_env.set("MXX_HIP_MATMUL_PATH", path)
    got = (gpu.GpuDCRTPolyMatrix.from_rns(p, a, True) * gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)).to_rns()
    assert np.array_equal(got, oracle.matmul(a, b, moduli))


@pytest.mark.parametrize("n,depth,bits,shape", [(4, 2, 17, (3, 5)), (2, 1, 17, (4, 3)), (128, 3, 24, (1, 6)), (128, 3, 24, (7, 1)),
                                                (1024, 2, 51, (5, 9)), (16384, 2, 24, (6, 4))])
def test_transpose_one_launch(gpu, oracle, n, depth, bits, shape):
    """gpupoly_matrix_transpose (one launch instead of rows*cols copy_block calls), including rings so small that a
    polynomial is shorter than 16 bytes and the vector shapes whose transpose is the same bytes."""
    moduli = oracle.gen_crt_basis(n, depth, bits)
    p = gpu.GpuDCRTPolyParams(n, moduli, 4)
    a = rand_matrix(oracle, 900, shape[0], shape[1], moduli, n)
    for fmt in (False, True):
        ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, fmt)
        t = ga.transpose()
        assert t.size() == (shape[1], shape[0]) and t.is_ntt == fmt
        assert np.array_equal(t.to_rns(), a.transpose(1, 0, 2, 3))
        assert t.transpose() == ga


@pytest.mark.parametrize("n,depth,bits,base,rows", [(64, 3, 17, 5, 1), (64, 3, 17, 5, 3), (32, 2, 51, 17, 1), (16384, 2, 24, 12, 1), (16384, 2, 24, 12, 2)])
def test_tensor_identity_extensions_equal_the_wrapper_loops(gpu, oracle, hip_env, n, depth, bits, base, rows):
    """gpupoly_matrix_mul_tensor_identity / _mul_tensor_identity_decompose / _mul_decompose_small against the
    reference's own loops (slice, per-column decompose, product, concat: gpu_dcrt_poly.rs:1374-1412,1495-1574), which
    the wrappers still run when MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH is set; row vectors take the in-place view path,
    taller operands the slice-buffer path."""
    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    k = p.modulus_digits()
    ks = -(-bits // base)
    ident, rows_b, cols_b = 3, 2, 3
    B = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 301, rows_b, cols_b, moduli, n), True)
    S = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 302, rows, rows_b * ident, moduli, n), True)
    Sd = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 303, rows, rows_b * k * ident, moduli, n), True)
    Ss = gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 304, rows, rows_b * ks, moduli, n), True)
    got = (S.mul_tensor_identity(B, ident), Sd.mul_tensor_identity_decompose(B, ident), Ss.mul_decompose_small(B))
    hip_env.set("MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH", "2")
    want = (S.mul_tensor_identity(B, ident), Sd.mul_tensor_identity_decompose(B, ident), Ss.mul_decompose_small(B))
    for g, w_ in zip(got, want):
        assert g.size() == w_.size() and g.is_ntt and g == w_
    assert got[0].size() == (rows, cols_b * ident)


@pytest.mark.parametrize("n,depth,bits", [(4, 2, 17), (128, 3, 24), (64, 2, 51), (16384, 2, 24)])
def test_device_side_constants_and_kronecker_product(gpu, oracle, n, depth, bits):
    """gpupoly_matrix_fill_zero / _fill_identity / _tensor against the host-built matrices the reference's wrappers
    upload (gpu_dcrt_poly.rs:343-365, 1158-1188) and the per-entry mul_scalar loop (:1225-1252)."""
    p = make_params(gpu, oracle, n, depth, bits, 4)
    moduli = p.moduli()
    L = len(moduli)
    for coeff_tag in (True, False):
        z = gpu.GpuDCRTPolyMatrix._new_zero_with_state(p, 3, 2, L - 1, coeff_tag)
        assert z.is_ntt == coeff_tag and z.size() == (3, 2) and not z.to_rns().any()
    ident = gpu.GpuDCRTPolyMatrix.identity(p, 3)
    want = np.zeros((3, 3, L, n), dtype=np.uint64)
    for i in range(3):
        want[i, i] = 1
    assert ident.is_ntt and np.array_equal(ident.to_rns(), want)
    s = rand_matrix(oracle, 410, 1, 1, moduli, n)
    gs = gpu.GpuDCRTPolyMatrix.from_rns(p, s, False)  # coefficient-domain scalar: identity() transforms it first
    scaled = gpu.GpuDCRTPolyMatrix.identity(p, 2, gpu.GpuDCRTPoly(gs))
    s_eval = oracle.matrix_ntt(s, moduli)
    want2 = np.zeros((2, 2, L, n), dtype=np.uint64)
    want2[0, 0] = want2[1, 1] = s_eval[0, 0]
    assert np.array_equal(scaled.to_rns(), want2)
    assert gpu.GpuDCRTPolyMatrix.identity(p, 0).size() == (0, 0)
    # Kronecker product: out[(i*rb + r), (j*cb + c)] = a[i][j] * b[r][c]
    a = oracle.matrix_ntt(rand_matrix(oracle, 411, 2, 3, moduli, n), moduli)
    b = oracle.matrix_ntt(rand_matrix(oracle, 412, 3, 2, moduli, n), moduli)
    t = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True).tensor(gpu.GpuDCRTPolyMatrix.from_rns(p, b, True))
    assert t.size() == (6, 6) and t.is_ntt
    got = t.to_rns()
    for i in range(2):
        for j in range(3):
            blk = oracle.pointwise("mul", np.broadcast_to(a[i, j], b.shape).copy(), b, moduli)
            assert np.array_equal(got[i * 3 : (i + 1) * 3, j * 2 : (j + 1) * 2], blk)
    # coefficient-domain operands: the wrapper transforms them, the result is the EVAL product (as in the reference)
    ac = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True).into_coeff_domain()
    bc = gpu.GpuDCRTPolyMatrix.from_rns(p, b, True).into_coeff_domain()
    assert ac.tensor(bc) == t


def test_add_rows_extension(gpu, oracle):
    """gpupoly_matrix_add_rows: out[r0 : r0 + rows] = a + b lands in a row block of a taller matrix and leaves the
    other rows alone (the preimage's final assembly uses it instead of copy_block + add_block)."""
    from mxx_amd import _ffi

    n = 256
    p = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p.moduli()
    base = rand_matrix(oracle, 500, 5, 3, moduli, n)
    a, b = rand_matrix(oracle, 501, 2, 3, moduli, n), rand_matrix(oracle, 502, 2, 3, moduli, n)
    out = gpu.GpuDCRTPolyMatrix.from_rns(p, base, True)
    out.add_rows_from(2, gpu.GpuDCRTPolyMatrix.from_rns(p, a, True), gpu.GpuDCRTPolyMatrix.from_rns(p, b, True))
    want = base.copy()
    want[2:4] = oracle.pointwise("add", a, b, moduli)
    assert np.array_equal(out.to_rns(), want)
    ga, gb = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True), gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
    with pytest.raises(gpu.GpuPolyError):  # rows 4..5 of a 5-row matrix
        _ffi.check_status(_ffi.lib().gpupoly_matrix_add_rows(out.raw, 4, ga.raw, gb.raw), "gpupoly_matrix_add_rows")


@pytest.mark.parametrize("n,depth,bits", [(16384, 3, 24), (16384, 2, 28), (256, 2, 24), (16384, 2, 51), (64, 2, 51)])
def test_ntt_add_rows_extension(gpu, oracle, n, depth, bits):
    """gpupoly_matrix_ntt_add_rows: out[r0 : r0 + rows] = NTT(coeff) + addend in one call - the fused 2^14 kernel
    (lazy and tight forms), and the copy / transform / add fall-back at other rings and with 64-bit words; the source
    stays as it was, the other rows of the destination too."""
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, n, depth, bits, 12)
    moduli = p.moduli()
    base = rand_matrix(oracle, 510, 4, 3, moduli, n)
    z, a = rand_matrix(oracle, 511, 2, 3, moduli, n), rand_matrix(oracle, 512, 2, 3, moduli, n)
    z[0, 0, :, :4] = 0
    z[1, 2] = (np.asarray(moduli, dtype=np.uint64) - 1)[:, None]  # extreme inputs
    a[1, 2] = (np.asarray(moduli, dtype=np.uint64) - 1)[:, None]
    out = gpu.GpuDCRTPolyMatrix.from_rns(p, base, True)
    gz, ga = gpu.GpuDCRTPolyMatrix.from_rns(p, z, False), gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
    out.ntt_add_rows_from(1, gz, ga)
    want = base.copy()
    want[1:3] = oracle.pointwise("add", oracle.matrix_ntt(z, moduli), a, moduli)
    assert out.is_ntt and np.array_equal(out.to_rns(), want)
    assert not gz.is_ntt and np.array_equal(gz.to_coeff_rns(), z)
    with pytest.raises(gpu.GpuPolyError):  # an EVAL source is refused, and so is a block past the last row
        _ffi.check_status(_ffi.lib().gpupoly_matrix_ntt_add_rows(out.raw, 1, ga.raw, ga.raw, 0), "gpupoly_matrix_ntt_add_rows")
    with pytest.raises(gpu.GpuPolyError):
        _ffi.check_status(_ffi.lib().gpupoly_matrix_ntt_add_rows(out.raw, 3, gz.raw, ga.raw, 0), "gpupoly_matrix_ntt_add_rows")
    # the consuming form (the source may be transformed in place where no fused kernel exists): same result
    out2 = gpu.GpuDCRTPolyMatrix.from_rns(p, base, True)
    out2.ntt_add_rows_from(1, gz, ga, consume=True)
    assert np.array_equal(out2.to_rns(), want)


def test_row_view_extension(gpu, oracle):
    """gpupoly_matrix_row_view: a row block as a matrix that shares its parent's storage - a product operand without
    the slice's copy; writes through the parent are seen by the view and the other way round; the parent outlives it."""
    import gc

    n = 256
    p = make_params(gpu, oracle, n, 2, 24, 12)
    moduli = p.moduli()
    m = rand_matrix(oracle, 520, 5, 3, moduli, n)
    left = rand_matrix(oracle, 521, 2, 3, moduli, n)
    gm = gpu.GpuDCRTPolyMatrix.from_rns(p, m, True)
    v = gm.row_view(1, 4)
    assert (v.row_size(), v.col_size()) == (3, 3) and np.array_equal(v.to_rns(), m[1:4])
    prod = gpu.GpuDCRTPolyMatrix.from_rns(p, left, True) * v
    assert np.array_equal(prod.to_rns(), oracle.matmul(left, m[1:4], moduli))
    gm.add_rows_from(2, gpu.GpuDCRTPolyMatrix.from_rns(p, m[0:1], True), gpu.GpuDCRTPolyMatrix.from_rns(p, m[4:5], True))
    assert np.array_equal(v.to_rns()[1], oracle.pointwise("add", m[0:1], m[4:5], moduli)[0])
    del gm  # the view keeps the storage alive
    gc.collect()
    assert np.array_equal(v.to_rns()[0], m[1])
    empty = gpu.GpuDCRTPolyMatrix.from_rns(p, m, True).row_view(5, 5)
    assert empty.row_size() == 0


@pytest.mark.parametrize("n,depth,bits", [(64, 3, 24), (256, 3, 51), (8, 2, 31)])
def test_mul_batch_extension(gpu, oracle, n, depth, bits):
    """gpupoly_matrix_mul_batch: a level's independent products in one call - mixed shapes, more than one launch's worth
    (70 > 64), 31-bit primes (accumulators folded inside the loop), 64-bit words; every product equals the single call."""
    p = make_params(gpu, oracle, n, depth, bits, 4)
    moduli = p.moduli()
    shapes = [(1, 5, 3), (2, 9, 4), (3, 1, 1), (1, 40, 2), (4, 4, 4)]
    lhss, rhss = [], []
    for i in range(70):
        r, k, c = shapes[i % len(shapes)]
        lhss.append(gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 600 + i, r, k, moduli, n), True))
        rhss.append(gpu.GpuDCRTPolyMatrix.from_rns(p, rand_matrix(oracle, 700 + i, k, c, moduli, n), i % 2 == 0))
    outs = gpu.GpuDCRTPolyMatrix.mul_batch(lhss, rhss)
    assert len(outs) == 70
    for l_, r_, o in zip(lhss, rhss, outs):
        assert o.is_ntt and o == l_ * r_.ensure_eval()
    # against the CPU restatement too (first few)
    for i in range(3):
        want = oracle.matmul(lhss[i].to_rns(), rhss[i].ensure_eval().to_rns(), moduli)
        assert np.array_equal(outs[i].to_rns(), want)
    assert gpu.GpuDCRTPolyMatrix.mul_batch([], []) == []


def test_mul_batch_large_products_and_errors(gpu, oracle):
    """products too large for the grouped launch run one by one through the tuned kernels; an output that is another
    product's operand is refused (the products of a batch are unordered)."""
    from mxx_amd import _ffi
    import ctypes as C

    p = make_params(gpu, oracle, 16384, 2, 24, 12)
    us = gpu.GpuDCRTPolyUniformSampler()
    a = [us.sample_uniform(p, 2, 6, gpu.DistType.FinRingDist()) for _ in range(3)]
    b = [us.sample_uniform(p, 6, 5, gpu.DistType.FinRingDist()) for _ in range(3)]
    for o, l_, r_ in zip(gpu.GpuDCRTPolyMatrix.mul_batch(a, b), a, b):
        assert o == l_ * r_
    sq = [us.sample_uniform(p, 2, 2, gpu.DistType.FinRingDist()) for _ in range(3)]
    arr = lambda ms: (C.c_void_p * len(ms))(*[m.raw.value for m in ms])
    st = _ffi.lib().gpupoly_matrix_mul_batch(arr([sq[2], sq[0]]), arr([sq[0], sq[1]]), arr([sq[1], sq[2]]), 2)
    assert st != 0 and "aliases" in _ffi.last_error_string()


@pytest.mark.parametrize("n,depth,bits", [(4, 2, 17), (256, 3, 24), (64, 2, 51)])
def test_neg_extension_and_three_operand_add_sub(gpu, oracle, n, depth, bits):
    """gpupoly_matrix_neg (one pass, also in place) and the wrapper's a + b / a - b through the three-operand ABI call
    into a fresh matrix, against the CPU restatement; zero residues stay zero under negation."""
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, n, depth, bits, 4)
    moduli = p.moduli()
    a, b = rand_matrix(oracle, 801, 2, 3, moduli, n), rand_matrix(oracle, 802, 2, 3, moduli, n)
    a[0, 0, :, :2] = 0
    zero = np.zeros_like(a)
    for fmt in (True, False):
        ga, gb = gpu.GpuDCRTPolyMatrix.from_rns(p, a, fmt), gpu.GpuDCRTPolyMatrix.from_rns(p, b, fmt)
        neg = -ga
        assert neg.is_ntt == fmt and np.array_equal(neg.to_rns(), oracle.pointwise("sub", zero, a, moduli))
        assert np.array_equal((ga + gb).to_rns(), oracle.pointwise("add", a, b, moduli))
        assert np.array_equal((ga - gb).to_rns(), oracle.pointwise("sub", a, b, moduli))
        assert np.array_equal(ga.to_rns(), a) and np.array_equal(gb.to_rns(), b)  # operands untouched
        _ffi.check_status(_ffi.lib().gpupoly_matrix_neg(ga.raw, ga.raw), "gpupoly_matrix_neg")  # in place
        assert ga == neg
    assert not (gpu.GpuDCRTPolyMatrix.from_rns(p, a, True) == gpu.GpuDCRTPolyMatrix.from_rns(p, b, True))


@pytest.mark.parametrize("n,depth,bits,base", [(256, 3, 51, 17), (1024, 2, 24, 12)])
def test_gate_batch_all_kinds_against_the_oracle(gpu, oracle, n, depth, bits, base):
    """gpupoly_batch: one level of independent gates - products, add / sub / negate, products by a ring element,
    decompositions - in one ABI call (src/circuit/poly_circuit/eval.rs:269-345 issues one call per gate); every output
    against the CPU restatement, including 70 point-wise gates (more than one launch's 64 descriptors), mixed shapes,
    COEFF operands for add / sub, and a gate at a lower level."""
    from mxx_amd import _ffi

    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    M = gpu.GpuDCRTPolyMatrix
    ev = lambda x: oracle.matrix_ntt(x, moduli)
    rnd = lambda seed, r, c: ev(rand_matrix(oracle, seed, r, c, moduli, n))
    gates, want = [], []
    for i in range(70):
        r, c = 1 + i % 3, 1 + (i // 3) % 4
        a, b = rnd(1000 + 2 * i, r, c), rnd(1001 + 2 * i, r, c)
        kind = ("add", "sub", "neg", "mul_scalar")[i % 4]
        if kind == "mul_scalar":
            s_ = rnd(1200 + i, 1, 1)
            gates.append((kind, M.from_rns(p, a, True), M.from_rns(p, s_, True)))
            want.append(oracle.pointwise("mul", a, np.broadcast_to(s_, a.shape).copy(), moduli))
        elif kind == "neg":
            gates.append((kind, M.from_rns(p, a, True), None))
            want.append(oracle.pointwise("sub", np.zeros_like(a), a, moduli))
        else:
            gates.append((kind, M.from_rns(p, a, i % 8 < 4), M.from_rns(p, b, i % 8 < 4)))  # EVAL and COEFF pairs
            want.append(oracle.pointwise(kind, a, b, moduli))
    for i in range(5):
        a, b = rnd(1400 + i, 2, 3 + i), rnd(1410 + i, 3 + i, 2)
        gates.append(("mul", M.from_rns(p, a, True), M.from_rns(p, b, True)))
        want.append(oracle.matmul(a, b, moduli))
    src = rand_matrix(oracle, 1500, 2, 2, moduli, n)
    gates.append(("decompose", M.from_rns(p, src, False), None))
    want.append(ev(oracle.decompose(src, moduli, base)))
    k = p.modulus_digits()
    sm, bm = rnd(1501, 1, 2 * k), rand_matrix(oracle, 1502, 2, 3, moduli, n)
    gates.append(("mul_decompose", M.from_rns(p, sm, True), M.from_rns(p, ev(bm), True)))
    want.append(oracle.matmul(sm, ev(oracle.decompose(bm, moduli, base)), moduli))
    # a gate at a lower level rides in the same call (its own launch: one limb count per launch)
    lo_a, lo_b = rnd(1600, 1, 2)[:, :, : depth - 1], rnd(1601, 1, 2)[:, :, : depth - 1]
    gates.append(("add", M.from_rns(p, np.ascontiguousarray(lo_a), True), M.from_rns(p, np.ascontiguousarray(lo_b), True)))
    want.append(oracle.pointwise("add", np.ascontiguousarray(lo_a), np.ascontiguousarray(lo_b), moduli[: depth - 1]))
    outs = M.eval_gates(gates)
    assert len(outs) == len(want)
    for i, (o, w, g) in enumerate(zip(outs, want, gates)):
        assert np.array_equal(o.to_rns(), w), (i, g[0])
        if g[0] in ("add", "sub"):
            assert o.is_ntt == g[1].is_ntt
    # aliasing between gates is refused: the gates of a level are unordered
    ops = (_ffi.GpuBatchOp * 2)()
    ops[0].kind, ops[0].out, ops[0].lhs, ops[0].rhs = _ffi.GPUPOLY_OP_ADD, outs[0].raw, outs[1].raw, outs[1].raw
    ops[1].kind, ops[1].out, ops[1].lhs, ops[1].rhs = _ffi.GPUPOLY_OP_ADD, outs[1].raw, outs[1].raw, outs[1].raw
    assert _ffi.lib().gpupoly_batch(ops, 2, base) != 0 and "aliases" in _ffi.last_error_string()


# ---- RNS snapshots, byte-slice transfers, stored matrix blocks (gpu_dcrt_poly.rs:72-120,576-711,1594-1641) ----------
def _snap_params(gpu):
    from mxx_amd.params import DCRTPolyParams

    moduli, _b, _d = DCRTPolyParams(256, 3, 20, 5).to_crt()
    return gpu.GpuDCRTPolyParams(256, moduli, 5)


def test_rns_snapshot_round_trip_and_validation(gpu):
    p = _snap_params(gpu)
    seed = gpu.GpuRngSeed.from_bytes(bytes(range(32)))
    for is_ntt in (True, False):
        m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 3, 2, gpu.GPU_MATRIX_DIST_UNIFORM, 0.0, seed)
        if not is_ntt:
            m = m.into_coeff_domain()
        snap = m.to_rns_snapshot()
        assert (snap.nrow(), snap.ncol(), snap.level(), snap.is_ntt()) == (3, 2, 2, is_ntt)
        assert snap.bytes_per_poly() == gpu.rns_bytes_len_for_level(p, 2) == 3 * 256 * 8 == gpu.rns_bytes_len(p)
        assert snap.bytes() == m.to_rns().tobytes()  # the numpy form of the same wire layout
        back = gpu.GpuDCRTPolyMatrix.from_rns_snapshot(p, snap)
        assert back.is_ntt == is_ntt and back == m
        assert back.to_rns_snapshot() == snap
        other = gpu.GpuDCRTPolyMatrix.zero(p, 3, 2) if is_ntt else gpu.GpuDCRTPolyMatrix.zero(p, 3, 2).into_coeff_domain()
        other.load_rns_snapshot(snap)
        assert other == m
    wrong_shape = gpu.GpuDCRTPolyMatrix.zero(p, 2, 3)
    with pytest.raises(AssertionError, match="row count mismatch"):
        wrong_shape.load_rns_snapshot(snap)
    wrong_format = gpu.GpuDCRTPolyMatrix.zero(p, 3, 2)  # EVAL, the last snapshot is COEFF
    with pytest.raises(AssertionError, match="format mismatch"):
        wrong_format.load_rns_snapshot(snap)
    bad = gpu.GpuDCRTMatrixRnsSnapshot(3, 2, 2, True, snap.bytes_per_poly(), snap.bytes()[:-8])
    with pytest.raises(AssertionError, match="byte length mismatch"):
        gpu.GpuDCRTPolyMatrix.from_rns_snapshot(p, bad)
    with pytest.raises(AssertionError, match="invalid RNS snapshot level"):
        gpu.GpuDCRTMatrixRnsSnapshot(1, 1, 3, True, 4 * 256 * 8, bytes(4 * 256 * 8)).validate_for_params(p)
    empty = gpu.GpuDCRTPolyMatrix.new_empty(p, 0, 4).to_rns_snapshot()
    assert empty.bytes() == b"" and gpu.GpuDCRTPolyMatrix.from_rns_snapshot(p, empty).size() == (0, 4)


def test_rns_bytes_with_a_padded_stride(gpu):
    """bytes_per_poly may exceed (level + 1) n 8 (a multiple of 8): the padding is skipped on load and left untouched on
    store; a store in the other format is refused with the reference's message (MatrixSerde.cu:765-768)."""
    p = _snap_params(gpu)
    seed = gpu.GpuRngSeed.from_bytes(bytes(range(1, 33)))
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 2, 2, gpu.GPU_MATRIX_DIST_UNIFORM, 0.0, seed)
    tight = m.bytes_per_poly()
    stride = tight + 64
    out = bytearray(b"\xa5" * (4 * stride))
    m.store_rns_bytes(out, stride, gpu.GPU_POLY_FORMAT_EVAL)
    wire = m.to_rns().tobytes()
    for i in range(4):
        assert out[i * stride : i * stride + tight] == wire[i * tight : (i + 1) * tight]
        assert out[i * stride + tight : (i + 1) * stride] == b"\xa5" * 64
    back = gpu.GpuDCRTPolyMatrix.new_empty(p, 2, 2)
    back.is_ntt = False
    back.load_rns_bytes(bytes(out), stride, gpu.GPU_POLY_FORMAT_EVAL)
    assert back.is_ntt and back == m
    with pytest.raises(RuntimeError, match="format conversion is not supported"):
        m.store_rns_bytes(out, stride, gpu.GPU_POLY_FORMAT_COEFF)
    m.store_rns_bytes(bytearray(), stride, gpu.GPU_POLY_FORMAT_EVAL)  # empty buffer: a no-op, as in the reference
    # the constant-coefficient store with a wider stride: words beyond the limb count stay as they were (MatrixSerde.cu:1041-1049)
    from mxx_amd import _ffi

    mc = m.into_coeff_domain()
    words = np.full((4, 5), 0xA5A5A5A5A5A5A5A5, dtype=np.uint64)
    ev = C.c_void_p()
    assert _ffi.lib().gpu_matrix_store_const_coeff_batch(mc.raw, words.ctypes.data, 5, C.byref(ev)) == 0
    _ffi.wait_and_destroy_events(ev)
    assert np.array_equal(words[:, :3], mc.store_const_coeff_words().reshape(4, 3)) and (words[:, 3:] == 0xA5A5A5A5A5A5A5A5).all()
    m = mc.ensure_eval()
    poly = m.entry(1, 0)
    buf = bytearray(tight)
    poly.store_rns_bytes(buf, gpu.GPU_POLY_FORMAT_EVAL)
    assert bytes(buf) == wire[2 * tight : 3 * tight]
    assert gpu.one_rns_bytes(p) == gpu.GpuDCRTPoly.const_one(p).inner.to_rns().tobytes()
    one = np.frombuffer(gpu.one_rns_bytes(p), dtype="<u8")
    assert one.size == 3 * 256 and (one == 1).all()  # the constant 1 evaluates to 1 in every slot of every limb


def test_read_from_files_blocks(gpu, tmp_path, monkeypatch):
    """a 5 x 3 matrix stored as BLOCK_SIZE = 2 blocks in the reference's file naming and bincode framing reads back
    entry for entry; a short entry is zero-padded and a missing file is an error naming it"""
    from mxx_amd.matrix import _bincode_nested_bytes

    p = _snap_params(gpu)
    seed = gpu.GpuRngSeed.from_bytes(bytes(range(2, 34)))
    m = gpu.GpuDCRTPolyMatrix.sample_distribution(p, 5, 3, gpu.GPU_MATRIX_DIST_UNIFORM, 0.0, seed)
    wire = m.to_rns()  # EVAL
    monkeypatch.setenv("BLOCK_SIZE", "2")
    bsize = 2
    ro, co = gpu.block_offsets(range(0, 5), bsize), gpu.block_offsets(range(0, 3), bsize)
    assert ro == [0, 2, 4, 5] and co == [0, 2, 3]
    for r0, r1 in zip(ro, ro[1:]):
        for c0, c1 in zip(co, co[1:]):
            entries = [[wire[i, j].tobytes() for j in range(c0, c1)] for i in range(r0, r1)]
            if (r0, c0) == (4, 2):
                entries[0][0] = entries[0][0][: 256 * 8]  # only limb 0 on file: the rest reads as zero
            (tmp_path / f"mat_{bsize}_{r0}.{r1}_{c0}.{c1}.matrix").write_bytes(_bincode_nested_bytes(entries))
    got = gpu.GpuDCRTPolyMatrix.read_from_files(p, 5, 3, tmp_path, "mat")
    want = wire.copy()
    want[4, 2, 1:] = 0
    assert got.is_ntt and np.array_equal(got.to_rns(), want)
    with pytest.raises(RuntimeError, match="Failed to read matrix file"):
        gpu.GpuDCRTPolyMatrix.read_from_files(p, 5, 3, tmp_path, "absent")


def test_poly_from_u64_vecs_sets_level_and_stays_coeff(gpu):
    p = _snap_params(gpu)
    moduli = p.moduli()
    coeffs = [[5, 6], [7], [], [9, 10]]  # two limbs at most -> level 1; missing residues are zero
    poly = gpu.GpuDCRTPoly.from_u64_vecs(p, coeffs)
    assert poly.level() == 1 and not poly.is_ntt()
    res = poly.inner.to_rns()[0, 0]
    assert res.shape == (2, 256)
    assert list(res[0, :5]) == [5, 7, 0, 9, 0] and list(res[1, :5]) == [6, 0, 0, 10, 0]
    full = gpu.GpuDCRTPoly.from_u64_vecs(p, [[1 % q for q in moduli]])
    assert full.level() == 2
    full.ntt_in_place()
    assert full == gpu.GpuDCRTPoly.const_one(p)
    with pytest.raises(AssertionError, match="exceeds CRT depth"):
        gpu.GpuDCRTPoly.from_u64_vecs(p, [[1, 2, 3, 4]])
    with pytest.raises(AssertionError, match="same level"):
        poly.assert_compatible(full)


def test_reference_named_constructors_and_perturbation_entry_points(gpu):
    """`new_with_gpu` / `reconstruct_coeffs_for_level` (gpu.rs:567-635), `sample_gpu_matrix_native` (sampler/gpu.rs:144)
    and the two perturbation entry points (trapdoor/gpu.rs:423-541) with a target width that is not a multiple of d"""
    import math

    from mxx_amd.params import DCRTPolyParams
    from mxx_amd.trapdoor import preimage_c, preimage_smoothing_parameter

    moduli, _b, _d = DCRTPolyParams(64, 3, 24, 12).to_crt()
    p = gpu.GpuDCRTPolyParams.new_with_gpu(64, moduli, 12, gpu.detected_gpu_device_ids()[:1])
    assert p == gpu.GpuDCRTPolyParams.new(64, moduli, 12) and p.crt_depth() == 3
    for level in range(3):
        Q = p.modulus_for_level(level)
        w = p.reconstruct_coeffs_for_level(level)
        assert len(w) == level + 1
        for i, wi in enumerate(w):  # the CRT idempotents: 1 mod q_i, 0 mod every other limb of the level
            assert 0 <= wi < Q and all(wi % q == (1 if j == i else 0) for j, q in enumerate(moduli[: level + 1]))
    x = 123456789012345 % p.modulus()
    poly = gpu.GpuDCRTPoly.from_biguint_to_constant(p, x)
    res = poly.ensure_coeff_domain().inner.to_rns()[0, 0, :, 0]
    assert sum(int(r) * wi for r, wi in zip(res, p.reconstruct_coeffs_for_level(2))) % p.modulus() == x

    a = gpu.sample_gpu_matrix_native(p, 2, 3, gpu.DistType.BitDist())
    b = gpu.sample_gpu_matrix_native(p, 2, 3, gpu.DistType.BitDist())
    assert a.size() == (2, 3) and a != b  # a fresh seed per call
    assert gpu.sample_gpu_matrix_native(p, 0, 3, gpu.DistType.BitDist()).size() == (0, 3)

    sigma, d, cols = 4.578, 2, 3  # 3 target columns, d = 2: the last block is padded to 4 and cut again
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, sigma)
    td, _pub = sampler.trapdoor(p, d)
    n, k, base = p.ring_dimension(), p.modulus_digits(), 1 << p.base_bits()
    c = preimage_c(base, sigma)
    s = preimage_smoothing_parameter(base, sigma, d, n, k)
    large = math.sqrt(s * s - c * c)
    from mxx_amd.sampler import seed_source

    with seed_source([bytes([i] * 32) for i in (1, 2)]):
        parts = sampler.sample_pert_square_mat_gpu_native_parts(p, td, s, c, sigma, large, cols)
    assert parts.p1.size() == (2 * d, 4) and parts.p2.size() == (d * k, 4)
    with seed_source([bytes([i] * 32) for i in (1, 2)]):
        p_hat = sampler.sample_pert_square_mat_gpu_native(p, td, s, c, sigma, large, cols)
    assert p_hat.size() == (2 * d + d * k, cols) and p_hat.is_ntt
    assert p_hat.slice(0, 2 * d, 0, cols) == parts.p1.slice_columns(0, cols)
    assert p_hat.slice(2 * d, 2 * d + d * k, 0, cols) == parts.p2.slice_columns(0, cols)
    assert td.get_or_create_p1_covariance_cache(c, s, sigma) is td.p1_covariance_cache(c, s, sigma)
    assert gpu.coeff_cached_matrix(td.r * td.r.transpose()) == td.a_mat_coeff
