"""GPU: the multi-GPU exchange step through the C ABI (gpupoly_comm_* / gpupoly_matrix_all_gather_columns) -
one process, a context per device, as the reference runs (`params_for_device`, src/poly/dcrt/gpu.rs:531-557;
`preimage_batched_sharded`, src/sampler/trapdoor/gpu.rs:371-397).  This box has ONE device: the RCCL backend is
exercised with a 1-device communicator, the shard logic with several contexts on that device (event-ordered peer
pulls - RCCL refuses two ranks per device).  Everything is compared with the oracle's concatenation."""
import numpy as np
import pytest

from conftest import make_params, rand_matrix

pytestmark = pytest.mark.gpu


def contexts_on_one_device(gpu, oracle, count, n=1024, depth=3, bits=24):
    p0 = make_params(gpu, oracle, n, depth, bits, 12)
    out = [p0]
    for i in range(1, count):  # same ring, different dnum -> its own context / stream / allocator cache
        out.append(gpu.GpuDCRTPolyParams(n, p0.moduli(), 12, gpu_ids=p0.gpu_ids(), dnum=100 + i))
    assert len({p.ctx_raw().value for p in out}) == count
    return out


@pytest.mark.parametrize("rows,shards,eval_format", [(1, [2], True), (3, [4], False), (1, [0], True)])
def test_all_gather_columns_rccl_one_device(gpu, oracle, rows, shards, eval_format):
    """ncclCommInitAll over one device, ncclAllGather on the context's stream, straight into the output (1 row) or
    through the padded staging block (3 rows); COEFF blocks keep their tag on both sides of the ABI."""
    from mxx_amd.parallel import GpuComm

    (p,) = contexts_on_one_device(gpu, oracle, 1)
    comm = GpuComm([p])
    assert comm.backend == "rccl" and len(comm) == 1
    moduli, n = p.moduli(), p.ring_dimension()
    x = np.ascontiguousarray(rand_matrix(oracle, 900 + rows, rows, max(shards[0], 1), moduli, n)[:, : shards[0]])
    local = gpu.GpuDCRTPolyMatrix.from_rns(p, x, eval_format)
    if shards[0]:
        local.ntt_all_in_place() if not eval_format else local.intt_all_in_place()  # work in flight on the stream
        local.ntt_all_in_place() if eval_format else local.intt_all_in_place()
    (full,) = comm.all_gather_columns([local])
    del local  # stream-ordered free right after the call
    assert full.is_ntt == eval_format and full.ncol == shards[0]
    assert np.array_equal(full.to_rns(), x)
    if shards[0]:  # the C-side tag travelled too: a format-checked entry point agrees with the Python flag
        other = gpu.GpuDCRTPolyMatrix.from_rns(p, x, eval_format)
        assert full == other
    comm.close()


@pytest.mark.parametrize("copy_form", ["runtime", "kernel"])
@pytest.mark.parametrize("rows,shards,eval_format", [(1, [2, 2], True), (1, [3, 1], False), (3, [2, 3, 1], True),
                                                      (2, [0, 4, 2], False), (22, [4, 3], True)])
def test_all_gather_columns_contexts_sharing_a_device(gpu, oracle, monkeypatch, rows, shards, eval_format, copy_form):
    """Several contexts (streams, allocators) on the one device: even, uneven and empty shards, one and many rows, both
    formats; the pull runs as runtime 2-D copies and as the copy kernel that crosses xGMI on a multi-GPU node."""
    from mxx_amd.parallel import GpuComm

    monkeypatch.setenv("MXX_HIP_COMM_COPY", copy_form)
    ps = contexts_on_one_device(gpu, oracle, len(shards))
    comm = GpuComm(ps)
    assert comm.backend == "peer"
    moduli, n = ps[0].moduli(), ps[0].ring_dimension()
    total = sum(shards)
    x = rand_matrix(oracle, 950 + rows + total, rows, total, moduli, n)
    want = oracle.matrix_ntt(x, moduli) if eval_format else x
    blocks, start = [], 0
    for p, c in zip(ps, shards):
        b = gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(x[:, start:start + c]), False)
        if eval_format:
            b.ntt_all_in_place()  # still in flight on this context's stream when the gather is enqueued
        blocks.append(b)
        start += c
    fulls = comm.all_gather_columns(blocks)
    from mxx_amd import _ffi

    for b in blocks:  # a block may be overwritten right after the call: the readers are ordered before it
        _ffi.check_status(_ffi.lib().gpupoly_matrix_fill_zero(b.raw), "gpupoly_matrix_fill_zero")
    del blocks
    for p, f in zip(ps, fulls):
        assert f.params is p and f.is_ntt == eval_format
        assert np.array_equal(f.to_rns(), want)
    comm.close()


def test_all_gather_columns_eight_contexts_bench_partitions(gpu, oracle):
    """The partitions an 8-GPU node sees, with eight contexts on the one device: the preimage's 22 x 50 matrix split
    7,7,6,6,6,6,6,6 (uneven, staged / 2-D copies) and the product's 1 x 120 split 15 x 8 (equal, straight into place)."""
    from mxx_amd.parallel import GpuComm, all_shard_ranges

    ps = contexts_on_one_device(gpu, oracle, 8, n=256, depth=2)
    comm = GpuComm(ps)
    moduli, n = ps[0].moduli(), 256
    for rows, cols in ((22, 50), (1, 120)):
        x = rand_matrix(oracle, 990 + rows, rows, cols, moduli, n)
        ranges = all_shard_ranges(cols, 8)
        assert [len(r) for r in ranges] == ([7, 7, 6, 6, 6, 6, 6, 6] if cols == 50 else [15] * 8)
        blocks = [gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(x[:, r.start:r.stop]), True) for p, r in zip(ps, ranges)]
        fulls = comm.all_gather_columns(blocks)
        for f in fulls:
            assert np.array_equal(f.to_rns(), x)
    comm.close()


def test_all_gather_columns_rejects_bad_arguments(gpu, oracle):
    from mxx_amd import _ffi
    from mxx_amd.parallel import GpuComm

    ps = contexts_on_one_device(gpu, oracle, 2)
    comm = GpuComm(ps)
    M = gpu.GpuDCRTPolyMatrix
    a, b = M(ps[0], 1, 2, 2, True), M(ps[1], 1, 2, 2, False)
    with pytest.raises(_ffi.GpuPolyError, match="different formats"):
        comm.all_gather_columns([a, b])
    with pytest.raises(_ffi.GpuPolyError, match="context r"):
        comm.all_gather_columns([a, M(ps[0], 1, 2, 2, True)])
    with pytest.raises(_ffi.GpuPolyError, match="sum of the blocks"):
        comm.all_gather_columns([a, M(ps[1], 1, 2, 2, True)], fulls=[M(ps[0], 1, 3, 2, True), M(ps[1], 1, 4, 2, True)])
    with pytest.raises(_ffi.GpuPolyError, match="duplicate context"):
        GpuComm([ps[0], ps[0]])
    other_ring = make_params(gpu, oracle, 1024, 2, 24, 12)
    with pytest.raises(_ffi.GpuPolyError, match="different rings"):
        GpuComm([ps[0], other_ring])
    comm.close()


def test_sharded_product_and_preimage_gathered_in_process(gpu, oracle):
    """The whole sharded path without torch: B / C column blocks and target columns split by shard_range over two
    contexts, one worker thread per context (the ABI calls release the GIL), one all-gather through the ABI; the
    gathered product equals the oracle's, the gathered preimage satisfies A x = u."""
    from concurrent.futures import ThreadPoolExecutor

    from mxx_amd.parallel import GpuComm, all_shard_ranges

    ps = contexts_on_one_device(gpu, oracle, 2, n=256, depth=2)
    comm = GpuComm(ps)
    moduli, n = ps[0].moduli(), 256
    a = oracle.matrix_ntt(rand_matrix(oracle, 41, 2, 3, moduli, n), moduli)
    b = oracle.matrix_ntt(rand_matrix(oracle, 42, 3, 5, moduli, n), moduli)
    ranges = all_shard_ranges(5, 2)

    def work(rank):
        p, sr = ps[rank], ranges[rank]
        ga = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True)
        gb = gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(b[:, sr.start:sr.stop]), True)
        return ga * gb

    with ThreadPoolExecutor(2) as ex:
        blocks = list(ex.map(work, range(2)))
    fulls = comm.all_gather_columns(blocks)
    want = oracle.matmul(a, b, moduli)
    for f in fulls:
        assert np.array_equal(f.to_rns(), want)
    # preimage: trapdoor replicated with copy_to_context, target columns sharded
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(ps[0], 4.578)
    td0, a0 = sampler.trapdoor(ps[0], 1)
    tds, pubs = [td0, td0.to_params(ps[1])], [a0, a0.to_params(ps[1])]
    target = gpu.GpuDCRTPolyUniformSampler().sample_uniform(ps[0], 1, 5, gpu.DistType.FinRingDist())
    t_rns = target.to_rns()

    def pre(rank):
        p, sr = ps[rank], ranges[rank]
        t = gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(t_rns[:, sr.start:sr.stop]), True)
        return sampler.preimage(p, tds[rank], pubs[rank], t)

    with ThreadPoolExecutor(2) as ex:
        xs = list(ex.map(pre, range(2)))
    fulls = comm.all_gather_columns(xs)
    for pub, f in zip(pubs, fulls):
        assert f.ncol == 5 and pub * f == gpu.GpuDCRTPolyMatrix.from_rns(f.params, t_rns, True)
    # the same as one call of the host mirror (an empty shard included)
    comm3 = GpuComm(ps + [gpu.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=ps[0].gpu_ids(), dnum=177)])
    p2 = comm3.params[2]
    cuts = [(0, 3), (3, 5), (5, 5)]
    shards = [(p, td0 if p is ps[0] else td0.to_params(p), a0 if p is ps[0] else a0.to_params(p),
               gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(t_rns[:, lo:hi]), True)) for p, (lo, hi) in zip(comm3.params, cuts)]
    fulls = sampler.preimage_column_sharded(comm3, shards)
    assert len(fulls) == 3 and fulls[2].params is p2
    for (p, _, a, _), f in zip(shards, fulls):
        assert f.ncol == 5 and a * f == gpu.GpuDCRTPolyMatrix.from_rns(p, t_rns, True)
    comm3.close()
    comm.close()


def test_gathered_blocks_self_validation_and_shifted_offset(gpu, oracle):
    """bench.py's self-validation on real gathered matrices (three contexts, uneven shards): every context checks EVERY
    block against a recomputation; a gather whose columns are rotated by one - what a wrong peer offset produces - is
    reported in every block, and a foreign block zeroed after the gather in exactly that block."""
    from mxx_amd.parallel import GpuComm, all_shard_ranges, blocks_that_differ

    ps = contexts_on_one_device(gpu, oracle, 3, n=256, depth=2)
    comm = GpuComm(ps)
    moduli, n = ps[0].moduli(), 256
    a = oracle.matrix_ntt(rand_matrix(oracle, 61, 1, 4, moduli, n), moduli)
    b = oracle.matrix_ntt(rand_matrix(oracle, 62, 4, 8, moduli, n), moduli)
    ranges = all_shard_ranges(8, 3)
    blocks = [gpu.GpuDCRTPolyMatrix.from_rns(p, a, True) * gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(b[:, r.start:r.stop]), True)
              for p, r in zip(ps, ranges)]
    fulls = comm.all_gather_columns(blocks)
    for p, f in zip(ps, fulls):
        whole = gpu.GpuDCRTPolyMatrix.from_rns(p, a, True) * gpu.GpuDCRTPolyMatrix.from_rns(p, b, True)
        eq = lambda m: (lambda lo, hi: m.slice_columns(lo, hi) == whole.slice_columns(lo, hi))
        assert blocks_that_differ(8, 3, eq(f)) == []
        shifted = f.slice_columns(1, 8).concat_columns([f.slice_columns(0, 1)])
        assert blocks_that_differ(8, 3, eq(shifted)) == [0, 1, 2]
        hole = f.clone()
        hole.copy_block_from(gpu.GpuDCRTPolyMatrix.zero(p, 1, len(ranges[1])), 0, ranges[1].start, 0, 0, 1, len(ranges[1]))
        assert blocks_that_differ(8, 3, eq(hole)) == [1]
    comm.close()


def test_bench_inproc_two_contexts_validates_foreign_blocks_and_fails_on_a_shift(gpu):
    """`bench.py --gpus 2 --inproc` (two contexts sharing this box's device): the line reports the exchange - ranks seen,
    backend, peer access, foreign blocks checked by every rank - and the same run with the gathered matrix rotated by one
    column (MXX_BENCH_FAULT_INJECT=shift) exits non-zero."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MXX_BENCH_INPROC_SHARE_DEVICES="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--inproc", "--steps", "2", "--warmup", "1", "--repeats", "0",
           "--no-trace"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    stdout_lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(stdout_lines) == 1 and len(stdout_lines[0]) < 4096  # the driver-facing line; the full record goes to stderr / bench_detail.json
    short = json.loads(stdout_lines[0])
    assert short["exchange"] == {"ranks_seen": 2, "comm_backend": "peer", "self_validated": True, "distinct_devices": False,
                                 "comm_verified_by_this_run": False, "preimage_self_validated": True}
    line = json.loads([l for l in out.stderr.splitlines() if l.startswith("bench.py detail: ")][-1][len("bench.py detail: "):])
    for blk in (line, line["preimage"]):
        ex = blk["exchange"]
        assert ex["ranks_seen"] == 2 and ex["comm_backend"] == "peer" and ex["self_validated"] is True
        assert ex["foreign_blocks_checked_per_rank"] == [1, 1] and ex["distinct_devices"] is False
    bad = subprocess.run(cmd + ["--workload", "m2a"], env=dict(env, MXX_BENCH_FAULT_INJECT="shift"), capture_output=True, text=True, timeout=900)
    assert bad.returncode != 0 and "differs from the recomputed one" in bad.stderr


@pytest.mark.skipif("__import__('mxx_amd').detected_gpu_device_count() < 2")
@pytest.mark.parametrize("backend", ["rccl", "peer"])
@pytest.mark.parametrize("rows,shards", [(1, [2, 2]), (1, [3, 1]), (3, [2, 3]), (2, [0, 4]), (22, [4, 3])])
def test_all_gather_columns_two_physical_devices(gpu, oracle, monkeypatch, backend, rows, shards):
    """Runs only where two devices are visible (never on this pool's one-GPU boxes - ADVICE r3): the RCCL communicator
    over distinct devices and the xGMI branches of the peer backend, even / uneven / empty / multi-row shards, against
    the oracle's concatenation."""
    from mxx_amd.parallel import GpuComm

    monkeypatch.setenv("MXX_HIP_COMM", backend)
    n, depth = 1024, 3
    moduli = oracle.gen_crt_basis(n, depth, 24)
    ps = [gpu.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=[dev]) for dev in range(len(shards))]
    comm = GpuComm(ps)
    assert comm.backend == backend
    total = sum(shards)
    x = rand_matrix(oracle, 1200 + rows + total, rows, total, moduli, n)
    want = oracle.matrix_ntt(x, moduli)
    blocks, start = [], 0
    for p, c in zip(ps, shards):
        b = gpu.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(x[:, start:start + c]), False)
        b.ntt_all_in_place()
        blocks.append(b)
        start += c
    fulls = comm.all_gather_columns(blocks)
    del blocks
    for p, f in zip(ps, fulls):
        assert f.params is p and np.array_equal(f.to_rns(), want)
    comm.close()


TWO_DEVICES = pytest.mark.skipif("__import__('mxx_amd').detected_gpu_device_count() < 2")


@TWO_DEVICES
@pytest.mark.parametrize("n,depth,bits,shape,eval_format", [(1024, 3, 24, (3, 5), True), (1024, 3, 24, (1, 7), False), (256, 3, 51, (2, 4), True),
                                                            (16384, 2, 24, (2, 3), True)])
def test_replica_on_a_second_physical_device(gpu, oracle, n, depth, bits, shape, eval_format):
    """SURVEY 8 row f3 across devices (`gpupoly_matrix_copy_to_context`: one peer copy over xGMI instead of the reference's
    to_cpu_staging_bytes -> from_cpu_staging_bytes round trip, src/lookup/ggh15/pubkey_gpu.rs:153-196).  Runs only where two
    devices are visible: the replica carries the same residues and the same format tag, lives on the other device, and
    arithmetic on it gives the oracle's result."""
    moduli = oracle.gen_crt_basis(n, depth, bits)
    p0, p1 = (gpu.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=[dev]) for dev in (0, 1))
    assert p0.ctx().device() == 0 and p1.ctx().device() == 1
    x = rand_matrix(oracle, 1500 + n, shape[0], shape[1], moduli, n)
    m0 = gpu.GpuDCRTPolyMatrix.from_rns(p0, x, eval_format)
    m1 = m0.to_params(p1)
    assert m1.params is p1 and m1.is_ntt == eval_format and np.array_equal(m1.to_rns(), x)
    back = m1.to_params(p0)
    assert back == m0
    if eval_format:
        y = rand_matrix(oracle, 1600 + n, shape[1], 2, moduli, n)
        prod = m1 * gpu.GpuDCRTPolyMatrix.from_rns(p1, y, True)
        assert np.array_equal(prod.to_rns(), oracle.matmul(x, y, moduli))


@TWO_DEVICES
def test_trapdoor_replica_and_column_sharded_preimage_two_physical_devices(gpu, oracle):
    """The reference's multi-device preimage (`preimage_batched_sharded`, src/sampler/trapdoor/gpu.rs:371-397; per-device
    params as src/poly/dcrt/gpu.rs:531-557) on two real devices: the trapdoor is replicated with peer copies, each device
    samples its column block, ONE all-gather through the C ABI leaves the whole preimage on both, and both satisfy A x = u;
    then the request fan-out itself: requests naming different devices run concurrently and come back in request order."""
    from mxx_amd.parallel import GpuComm

    n, depth = 1024, 3
    moduli = oracle.gen_crt_basis(n, depth, 24)
    ps = [gpu.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=[dev]) for dev in (0, 1)]
    sampler = gpu.GpuDCRTPolyTrapdoorSampler(ps[0], 4.578)
    td0, a0 = sampler.trapdoor(ps[0], 1)
    td1, a1 = td0.to_params(ps[1]), a0.to_params(ps[1])
    assert td1.r.params is ps[1] and np.array_equal(td1.r.to_rns(), td0.r.to_rns()) and np.array_equal(a1.to_rns(), a0.to_rns())
    t_rns = oracle.matrix_ntt(rand_matrix(oracle, 1700, 1, 7, moduli, n), moduli)
    comm = GpuComm(ps)
    shards = [(ps[0], td0, a0, gpu.GpuDCRTPolyMatrix.from_rns(ps[0], np.ascontiguousarray(t_rns[:, :4]), True)),
              (ps[1], td1, a1, gpu.GpuDCRTPolyMatrix.from_rns(ps[1], np.ascontiguousarray(t_rns[:, 4:]), True))]
    fulls = sampler.preimage_column_sharded(comm, shards)
    assert [f.params for f in fulls] == ps
    for (p, _, a, _), f in zip(shards, fulls):
        assert f.ncol == 7 and a * f == gpu.GpuDCRTPolyMatrix.from_rns(p, t_rns, True)
    assert np.array_equal(fulls[0].to_rns(), fulls[1].to_rns())
    comm.close()
    reqs = []
    for j in range(6):
        dev = j % 2
        t = gpu.GpuDCRTPolyMatrix.from_rns(ps[dev], np.ascontiguousarray(t_rns[:, j:j + 1]), True)
        reqs.append((40 + j, ps[dev], (td0, td1)[dev], (a0, a1)[dev], t))
    out = sampler.preimage_batched_sharded(reqs)
    assert [i for i, _ in out] == [40 + j for j in range(6)]
    for (_, x), (_, p, _, a, t) in zip(out, reqs):
        assert x.params is p and a * x == t
