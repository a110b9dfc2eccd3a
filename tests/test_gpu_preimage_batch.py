"""GPU parity: several preimage requests in ONE sequence of launches (`preimage_many`, the `gpupoly_*_segments` entry points,
`gpupoly_matrix_concat_columns` / `_split_columns`), the reference's own call sequence, and the trapdoor's public-matrix cache.

The bar for batching is the strongest one available: a request's output must be the matrix it would get ALONE - bit for bit,
for the same seeds - and that matrix is itself checked against the CPU restatement (oracle.preimage) request by request.
"""
import numpy as np
import pytest

from conftest import make_params

pytestmark = pytest.mark.gpu

SIGMA = 4.578


def seed_bytes(tag):
    return bytes((tag * 29 + 7 * i + 3) & 0xFF for i in range(32))


def gseed(gpu, tag):
    return gpu.GpuRngSeed.from_bytes(seed_bytes(tag))


# ---------------------------------------------------------------------------------------------------
# the three segmented samplers against the plain entry points and the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,depth,bits,base,rows,seg_cols", [
    (256, 3, 51, 17, 3, [4, 1, 2, 5]),      # u64 words, the GGH15 chain's ring
    (1024, 2, 24, 12, 2, [2, 3]),           # u32 words
    (128, 2, 24, 12, 1, [1] * 70),          # more segments than one launch's table holds on the host side: the mirror chunks
    (16384, 2, 24, 12, 1, [1, 2]),          # several elements per lane
])
def test_gaussian_segments_equal_the_plain_sampler_per_segment(gpu, oracle, n, depth, bits, base, rows, seg_cols):
    p = make_params(gpu, oracle, n, depth, bits, base)
    M, moduli = gpu.GpuDCRTPolyMatrix, p.moduli()
    sigma, code = 1234.5, gpu.DistType.GaussDist(1234.5).as_ffi()
    got_parts = []
    for lo in range(0, len(seg_cols), 64):
        chunk = seg_cols[lo:lo + 64]
        m = M.sample_distribution_segments(p, rows, chunk, code, sigma, [gseed(gpu, 100 + lo + j) for j in range(len(chunk))])
        assert m.is_ntt and m.size() == (rows, sum(chunk))
        got_parts += m.split_columns(chunk)
    for j, (c, part) in enumerate(zip(seg_cols, got_parts)):
        alone = M.sample_distribution(p, rows, c, code, sigma, gseed(gpu, 100 + j))
        assert part == alone, f"segment {j} differs from the plain sampler"
        if j < 3:
            want = oracle.matrix_ntt(oracle.sample_distribution(rows, c, moduli, n, "gauss", sigma, seed_bytes(100 + j)), moduli)
            assert np.array_equal(part.to_rns(), want)


@pytest.mark.parametrize("n,depth,bits,base,d,seg_cols", [(256, 3, 51, 17, 2, [4, 2, 6]), (1024, 2, 24, 12, 1, [3, 1, 1, 2]),
                                                          (256, 2, 24, 12, 2, [2] * 9)])
def test_p1_segments_equal_the_plain_sampler_per_segment(gpu, oracle, n, depth, bits, base, d, seg_cols):
    from mxx_amd.trapdoor import GpuDCRTTrapdoor, p1_covariance_parameters

    p = make_params(gpu, oracle, n, depth, bits, base)
    M, moduli = gpu.GpuDCRTPolyMatrix, p.moduli()
    from mxx_amd.sampler import seed_source

    with seed_source([seed_bytes(1), seed_bytes(2)]):
        td = GpuDCRTTrapdoor.new(p, d, SIGMA)
    c, s, dgg = p1_covariance_parameters(p, d, SIGMA)
    cache = td.p1_covariance_cache(c, s, dgg)
    total = sum(seg_cols)
    tp2 = M.sample_distribution(p, 2 * d, total, gpu.DistType.GaussDist(3.0e5).as_ffi(), 3.0e5, gseed(gpu, 9))
    seeds = [gseed(gpu, 40 + j) for j in range(len(seg_cols))]
    got = M.sample_p1_full_cached_segments(cache, tp2.clone(), seeds, seg_cols)
    assert got.is_ntt
    parts, tparts = got.split_columns(seg_cols), tp2.split_columns(seg_cols)
    for j, (part, t) in enumerate(zip(parts, tparts)):
        alone = M.sample_p1_full_cached(cache, t.clone(), seeds[j])
        assert part == alone, f"segment {j} differs from the plain p1 sampler"
    # one segment against the CPU restatement too
    inv = lambda m: oracle.matrix_ntt(m, moduli, inverse=True)
    r, e = td.r.to_rns(), td.e.to_rns()
    rt, et = np.swapaxes(r, 0, 1), np.swapaxes(e, 0, 1)
    sv, up = oracle.p1_covariance(inv(oracle.matmul(r, rt, moduli)), inv(oracle.matmul(r, et, moduli)), inv(oracle.matmul(e, et, moduli)),
                                  moduli, c, s, dgg)
    want = oracle.matrix_ntt(oracle.sample_p1(inv(tparts[1].to_rns()), moduli, sv, up, -(c * c) / (s * s - c * c), seed_bytes(41)), moduli)
    assert np.array_equal(parts[1].to_rns(), want)


@pytest.mark.parametrize("n,depth,bits,base,rows,seg_cols", [(256, 3, 51, 17, 2, [4, 1, 3]), (1024, 2, 24, 12, 1, [2, 2, 1]),
                                                             (256, 2, 24, 6, 2, [1, 5]),  # four digits per tower
                                                             (16384, 2, 24, 12, 1, [2, 1])])
def test_gadget_sampler_segments_equal_the_plain_sampler_per_segment(gpu, oracle, n, depth, bits, base, rows, seg_cols):
    p = make_params(gpu, oracle, n, depth, bits, base)
    M, moduli = gpu.GpuDCRTPolyMatrix, p.moduli()
    c = (2.0 ** base + 1.0) * SIGMA
    src = M.from_rns(p, oracle.random_matrix(77, rows, sum(seg_cols), moduli, n), False)
    seeds = [gseed(gpu, 60 + j) for j in range(len(seg_cols))]
    got = src.clone().gauss_samp_gq_arb_base_segments(c, SIGMA, seeds, seg_cols)
    assert got.is_ntt and got.size() == (rows * p.modulus_digits(), sum(seg_cols))
    coeff = src.clone().gauss_samp_gq_arb_base_segments(c, SIGMA, seeds, seg_cols, coeff_out=True)
    assert not coeff.is_ntt and coeff.ensure_eval() == got
    for j, (part, sp) in enumerate(zip(got.split_columns(seg_cols), src.split_columns(seg_cols))):
        want_coeff = oracle.gauss_samp_gq(sp.to_rns(), moduli, base, c, seed_bytes(60 + j))
        alone = sp.gauss_samp_gq_arb_base(c, SIGMA, seeds[j])
        assert part == alone, f"segment {j} differs from the plain G-sampler"
        assert np.array_equal(part.to_rns(), oracle.matrix_ntt(want_coeff, moduli))
    g = M.gadget_matrix(p, rows)
    assert g * got == src.ensure_eval()  # G z = u: the coset condition, whatever the segments


def test_segment_entry_points_refuse_what_they_do_not_cover(gpu, oracle):
    from mxx_amd._ffi import GpuPolyError

    M = gpu.GpuDCRTPolyMatrix
    small = make_params(gpu, oracle, 64, 2, 24, 12)
    with pytest.raises(GpuPolyError, match="unsupported"):
        M.sample_distribution_segments(small, 1, [1, 1], gpu.DistType.GaussDist(3.0).as_ffi(), 3.0, [gseed(gpu, 1), gseed(gpu, 2)])
    p = make_params(gpu, oracle, 256, 2, 24, 12)
    with pytest.raises(GpuPolyError, match="unsupported"):
        M.sample_distribution_segments(p, 1, [1, 1], gpu.DistType.FinRingDist().as_ffi(), 0.0, [gseed(gpu, 1), gseed(gpu, 2)])
    with pytest.raises(GpuPolyError, match="cover the matrix"):
        out = M.new_empty(p, 1, 3)
        from mxx_amd import _ffi
        import ctypes as C

        arr = (gpu.GpuRngSeed * 2)(gseed(gpu, 1), gseed(gpu, 2))
        cols = (C.c_size_t * 2)(1, 1)
        _ffi.check_status(_ffi.lib().gpupoly_matrix_sample_distribution_segments(out.raw, 1, 3.0, arr, cols, 2), "segments")
    deep = make_params(gpu, oracle, 256, 2, 24, 4)  # six digits per tower: the G-sampler's lane kernel stops at four
    src = M.from_rns(deep, oracle.random_matrix(5, 1, 2, deep.moduli(), 256), False)
    with pytest.raises(GpuPolyError, match="unsupported"):
        src.gauss_samp_gq_arb_base_segments(17.0 * SIGMA, SIGMA, [gseed(gpu, 1), gseed(gpu, 2)], [1, 1])


# ---------------------------------------------------------------------------------------------------
# column blocks
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,depth,bits", [(256, 3, 51), (1024, 2, 24), (2, 1, 17)])
def test_concat_and_split_columns_in_one_launch(gpu, oracle, n, depth, bits):
    p = make_params(gpu, oracle, n, depth, bits, 12)
    M, moduli = gpu.GpuDCRTPolyMatrix, p.moduli()
    widths = [3, 1, 4, 1, 5] + [1] * 70  # more than 64 blocks: two launches
    blocks_np = [oracle.random_matrix(900 + j, 3, w, moduli, n) for j, w in enumerate(widths)]
    blocks = [M.from_rns(p, b, True) for b in blocks_np]
    wide = M.concat_columns_of(blocks)
    assert wide.is_ntt and np.array_equal(wide.to_rns(), np.concatenate(blocks_np, axis=1))
    assert wide == blocks[0].concat_columns(blocks[1:])
    back = wide.split_columns(widths)
    for b, want in zip(back, blocks_np):
        assert b.is_ntt and np.array_equal(b.to_rns(), want)
    # an empty block takes no part; domains are unified like concat_columns does
    mixed = [blocks[0], M(p, 3, 0, depth - 1, True), blocks[1].clone().into_coeff_domain()]
    assert M.concat_columns_of(mixed) == blocks[0].concat_columns([blocks[1]])
    from mxx_amd._ffi import GpuPolyError

    with pytest.raises(GpuPolyError, match="add up"):
        from mxx_amd import _ffi

        _ffi.check_status(_ffi.lib().gpupoly_matrix_split_columns(wide.raw, M._raw_array(back[:2]), 2), "split")


# ---------------------------------------------------------------------------------------------------
# whole preimages
# ---------------------------------------------------------------------------------------------------
def trapdoor_and_oracle(gpu, oracle, p, n, base, d, master):
    from mxx_amd.sampler import seed_source

    moduli = p.moduli()
    r, e, a = oracle.trapdoor_gen(moduli, n, base, SIGMA, d, master)
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(p, SIGMA)
    with seed_source([oracle._seed_from(master, i).tobytes() for i in range(3)]):
        td, A = sampler.trapdoor(p, d)
    assert np.array_equal(A.to_rns(), a)
    return sampler, td, A, (r, e, a)


@pytest.mark.parametrize("n,depth,bits,base,d,cols", [
    (256, 12, 51, 17, 2, [4, 4, 4, 4, 4, 4, 4, 4]),   # the M4 request, eight at a time
    (256, 3, 51, 17, 2, [3, 1, 4, 2, 5]),             # odd column counts: the perturbation of a request is padded to a multiple of d
    (1024, 3, 24, 12, 1, [1, 2, 3, 1]),               # u32 words, d = 1
    (128, 2, 24, 12, 1, [1] * 67),                    # more requests than one launch's segment table
])
def test_batched_requests_equal_one_by_one_and_the_oracle(gpu, oracle, n, depth, bits, base, d, cols):
    from mxx_amd.sampler import seed_source

    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    sampler, td, A, (r, e, a) = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(5))
    M = gpu.GpuDCRTPolyMatrix
    masters = [seed_bytes(50 + j) for j in range(len(cols))]
    targets_np = [oracle.matrix_ntt(oracle.random_matrix(3000 + j, d, c, moduli, n), moduli) for j, c in enumerate(cols)]
    targets = [M.from_rns(p, t, True) for t in targets_np]
    draws = [oracle._seed_from(m, i).tobytes() for m in masters for i in (3, 4, 5)]  # request by request: p2, p1, z
    with seed_source(draws):
        batched = sampler.preimage_many(p, td, A, targets)
    with seed_source(draws):
        alone = [sampler.preimage(p, td, A, t) for t in targets]
    k = p.modulus_digits()
    for j, (xb, xa, t) in enumerate(zip(batched, alone, targets)):
        assert xb.size() == ((k + 2) * d, cols[j]) and xb.is_ntt
        assert xb == xa, f"request {j}: batched preimage differs from the one sampled alone"
        assert A * xb == t
    for j in range(min(3, len(cols))):  # the CPU restatement, request by request
        want = oracle.preimage(moduli, n, base, SIGMA, r, e, a, targets_np[j], masters[j])
        assert np.array_equal(batched[j].to_rns(), want)


def test_batching_saves_launches_and_keeps_request_order(gpu, oracle):
    from mxx_amd import _ffi
    from mxx_amd.sampler import seed_source

    n, depth, bits, base, d = 256, 12, 51, 17, 2
    p = make_params(gpu, oracle, n, depth, bits, base)
    sampler, td, A, _ = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(6))
    _, td2, A2, _ = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(7))
    us = gpu.GpuDCRTPolyUniformSampler()
    targets = [us.sample_uniform(p, d, 4, gpu.DistType.FinRingDist()) for _ in range(16)]
    lib = _ffi.lib()
    c0 = lib.gpupoly_launch_count()
    one = sampler.preimage(p, td, A, targets[0])
    c1 = lib.gpupoly_launch_count()
    many = sampler.preimage_many(p, td, A, targets)
    c2 = lib.gpupoly_launch_count()
    assert A * one == targets[0] and all(A * x == t for x, t in zip(many, targets))
    assert (c2 - c1) < 3 * (c1 - c0), f"16 requests took {c2 - c1} launches, one takes {c1 - c0}"
    # preimage_batched_sharded: mixed trapdoors in one call, results in request order with their entry indices
    reqs = []
    for j in range(10):
        use2 = j % 3 == 1
        reqs.append((100 + j, p, td2 if use2 else td, A2 if use2 else A, targets[j]))
    draws = [seed_bytes(200 + i) for i in range(30)]
    with seed_source(draws):
        out = sampler.preimage_batched_sharded(reqs)
    with seed_source(draws):
        want = [(idx, sampler.preimage(pp, t_, a_, u_)) for idx, pp, t_, a_, u_ in reqs]
    assert [i for i, _ in out] == [100 + j for j in range(10)]
    for (i, x), (_, w), (_, _, _, a_, u_) in zip(out, want, reqs):
        assert x == w and a_ * x == u_


def test_shapes_without_a_segmented_form_fall_back_to_one_by_one(gpu, oracle, monkeypatch):
    from mxx_amd.sampler import seed_source

    n, depth, bits, base, d = 64, 2, 24, 12, 1  # n below the segmented samplers' 128
    p = make_params(gpu, oracle, n, depth, bits, base)
    sampler, td, A, _ = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(8))
    assert not sampler._segments_supported(p, td)
    us = gpu.GpuDCRTPolyUniformSampler()
    targets = [us.sample_uniform(p, d, c, gpu.DistType.FinRingDist()) for c in (1, 2, 1)]
    draws = [seed_bytes(300 + i) for i in range(9)]
    with seed_source(draws):
        many = sampler.preimage_many(p, td, A, targets)
    with seed_source(draws):
        alone = [sampler.preimage(p, td, A, t) for t in targets]
    assert all(x == y for x, y in zip(many, alone))
    # MXX_PREIMAGE_BATCH=0 switches batching off where it exists
    p2 = make_params(gpu, oracle, 256, 2, 24, 12)
    s2, td2, A2, _ = trapdoor_and_oracle(gpu, oracle, p2, 256, 12, 1, seed_bytes(9))
    assert s2._segments_supported(p2, td2)
    monkeypatch.setenv("MXX_PREIMAGE_BATCH", "0")
    assert not s2._segments_supported(p2, td2)


@pytest.mark.parametrize("n,depth,bits,base,d,cols", [(256, 3, 51, 17, 2, 3), (1024, 3, 24, 12, 1, 5), (16384, 2, 24, 12, 1, 4)])
def test_reference_call_sequence_gives_the_same_preimage(gpu, oracle, n, depth, bits, base, d, cols, monkeypatch):
    """an unpatched mxx calls only the `gpu_*` entry points, in its own order: same seeds -> same residues as the extension
    sequence `preimage` uses (small- and large-operand assembly), and as the CPU restatement"""
    from mxx_amd import trapdoor as tdmod
    from mxx_amd.sampler import seed_source

    p = make_params(gpu, oracle, n, depth, bits, base)
    moduli = p.moduli()
    sampler, td, A, (r, e, a) = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(11))
    master = seed_bytes(12)
    t_np = oracle.matrix_ntt(oracle.random_matrix(81, d, cols, moduli, n), moduli)
    t = gpu.GpuDCRTPolyMatrix.from_rns(p, t_np, True)
    draws = [oracle._seed_from(master, i).tobytes() for i in (3, 4, 5)]
    with seed_source(draws):
        ref = sampler.preimage_reference_sequence(p, td, A, t)
    with seed_source(draws):
        ext = sampler.preimage(p, td, A, t)
    monkeypatch.setattr(tdmod, "TRAFFIC_BOUND_BYTES", 0)  # the large-operand assembly at this size
    with seed_source(draws):
        ext_large = sampler.preimage(p, td, A, t)
    assert ref == ext and ref == ext_large and A * ref == t
    assert np.array_equal(ref.to_rns(), oracle.preimage(moduli, n, base, SIGMA, r, e, a, t_np, master))


def test_public_matrix_cache_follows_in_place_writes(gpu, oracle):
    """the trapdoor keeps A's column blocks per public-matrix OBJECT; rewriting that object in place (copy_block_from,
    load_rns ...) must not leave stale blocks behind - preimage() has to solve for the NEW matrix"""
    n, depth, bits, base, d = 256, 2, 24, 12, 1
    p = make_params(gpu, oracle, n, depth, bits, base)
    sampler, td, A, _ = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(13))
    us = gpu.GpuDCRTPolyUniformSampler()
    u = us.sample_uniform(p, d, 2, gpu.DistType.FinRingDist())
    assert A * sampler.preimage(p, td, A, u) == u
    v0 = A.content_version()
    parts0 = td.public_matrix_parts(A)
    assert td.public_matrix_parts(A) is parts0  # cached while untouched
    # the same object now holds ANOTHER trapdoor's public matrix
    sampler2, td2, A2, _ = trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(14))
    A.copy_block_from(A2, 0, 0, 0, 0, A2.row_size(), A2.col_size())
    assert A.content_version() > v0 and A == A2
    assert td.public_matrix_parts(A) is not parts0
    x = sampler2.preimage(p, td2, A, u)
    assert A * x == u
    view = A.row_view(0, 1)
    v1 = A.content_version()
    view.add_in_place(view)  # a write through a view bumps the parent too
    assert A.content_version() > v1
    td.clear_public_matrix_cache()
    assert td._stacked is None


@pytest.mark.parametrize("workers", ["1", "3"])
def test_requests_with_different_trapdoors_overlap_on_worker_contexts(gpu, oracle, monkeypatch, workers):
    """one context, several trapdoors (the GGH15 levels): key groups are dealt to worker contexts on the same device (a
    stream and an allocator each, replicas of (trapdoor, A) by device-to-device copies); every request still gets the
    matrix it would get alone, in the context it named, in request order"""
    from mxx_amd import trapdoor as tdmod
    from mxx_amd.sampler import seed_source

    monkeypatch.setenv("MXX_PREIMAGE_WORKERS", workers)
    n, depth, bits, base, d = 256, 3, 51, 17, 2
    p = make_params(gpu, oracle, n, depth, bits, base)
    keys = [trapdoor_and_oracle(gpu, oracle, p, n, base, d, seed_bytes(20 + j)) for j in range(3)]
    sampler = keys[0][0]
    us = gpu.GpuDCRTPolyUniformSampler()
    reqs = []
    for j in range(7):
        _, td, A, _ = keys[j % 3]
        reqs.append((j, p, td, A, us.sample_uniform(p, d, 1 + j % 3, gpu.DistType.FinRingDist())))
    draws = [seed_bytes(400 + i) for i in range(21)]
    with seed_source(draws):
        out = sampler.preimage_batched_sharded(reqs)
    with seed_source(draws):
        want = [sampler.preimage(pp, t_, a_, u_) for _, pp, t_, a_, u_ in reqs]
    assert [i for i, _ in out] == list(range(7))
    for (_, x), w, (_, _, _, a_, u_) in zip(out, want, reqs):
        assert x.params.ctx_raw().value == p.ctx_raw().value and x == w and a_ * x == u_
    if workers != "1":
        pw = tdmod.worker_params(p, 1)
        assert pw.ctx_raw().value != p.ctx_raw().value and pw.ctx().device() == p.ctx().device()
        td1 = keys[1][1]
        assert pw.ctx_raw().value in td1._replicas  # the second key group ran on worker 1, on replicas
        r_w = td1._replicas[pw.ctx_raw().value][0].r
        assert r_w.params is pw and np.array_equal(r_w.to_rns(), td1.r.to_rns())
