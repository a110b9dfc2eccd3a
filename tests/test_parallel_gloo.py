"""CPU-only: the N>1 path (rank-wise sharding, barrier + max-over-ranks clock, all-gather of
output column blocks) with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

import json

import numpy as np

from mxx_amd.parallel import all_shard_ranges, padded_len, place_column_blocks, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_and_balance():
    for total, world in [(50, 8), (120, 8), (64, 8), (1024, 2), (7, 8), (0, 4), (50, 1)]:
        rs = all_shard_ranges(total, world)
        assert rs[0].start == 0 and rs[-1].stop == total
        assert all(a.stop == b.start for a, b in zip(rs, rs[1:]))
        sizes = [len(r) for r in rs]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == total
        assert padded_len(total, world) == (max(sizes) if total else 0)
    assert [len(r) for r in all_shard_ranges(50, 8)] == [7, 7, 6, 6, 6, 6, 6, 6]
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


WORKER = textwrap.dedent(
    """
    import os, sys, time
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch
    import torch.distributed as dist
    from mxx_amd.parallel import shard_range, all_gather_column_blocks, padded_len, place_column_blocks
    from oracle import oracle as O

    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # column-sharded ring-matrix product C = A * B with the oracle standing in for the device
    n, moduli = 16, O.gen_crt_basis(16, 2, 17)
    A = O.matrix_ntt(O.random_matrix(1, 2, 3, moduli, n), moduli)
    B = O.matrix_ntt(O.random_matrix(2, 3, 6, moduli, n), moduli)
    sr = shard_range(B.shape[1], world, rank)
    C_local = O.matmul(A, np.ascontiguousarray(B[:, sr.start:sr.stop]), moduli)   # rows x cols_local
    dist.barrier()
    t0 = time.perf_counter()
    blocks = all_gather_column_blocks(torch.from_numpy(C_local.reshape(-1).view(np.int64)), B.shape[1], dist, None, world)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    full = np.concatenate([b.numpy().view(np.uint64).reshape(2, -1, len(moduli), n) for b in blocks], axis=1)
    want = O.matmul(A, B, moduli)
    assert np.array_equal(full, want), "gathered product differs"
    # unequal shards (7 columns over 2 ranks -> 4 + 3): pad to the largest shard, gather, un-pad with the
    # same placement rule ColumnAllGather applies on the device (bench.py --scaling strong)
    B7 = O.matrix_ntt(O.random_matrix(3, 3, 7, moduli, n), moduli)
    s7 = shard_range(7, world, rank)
    pad = padded_len(7, world)
    block = np.zeros((2, pad, len(moduli), n), dtype=np.uint64)
    block[:, : len(s7)] = O.matmul(A, np.ascontiguousarray(B7[:, s7.start:s7.stop]), moduli)
    recv = torch.empty(world * block.size, dtype=torch.int64)
    dist.all_gather_into_tensor(recv, torch.from_numpy(block.reshape(-1).view(np.int64)))
    full7 = place_column_blocks(recv.numpy().view(np.uint64).reshape((world,) + block.shape), 7)
    assert np.array_equal(full7, O.matmul(A, B7, moduli)), "padded gather differs"
    # preimage-style column sharding: 50 target columns, every column owned exactly once
    owned = torch.zeros(50, dtype=torch.int64)
    s = shard_range(50, world, rank)
    owned[s.start:s.stop] = 1
    dist.all_reduce(owned)
    assert bool((owned == 1).all())
    if rank == 0:
        print("GLOO_OK", float(el.item()) >= 0.0)
    dist.destroy_process_group()
    """
)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "GLOO_OK True" in out.stdout


def test_place_column_blocks_unpads():
    world, rows, total = 3, 2, 8
    full = np.arange(rows * total * 5).reshape(rows, total, 5)
    pad = padded_len(total, world)
    g = np.full((world, rows, pad, 5), -1)
    for r, sr in enumerate(all_shard_ranges(total, world)):
        g[r, :, : len(sr)] = full[:, sr.start : sr.stop]
    assert np.array_equal(place_column_blocks(g, total), full)


def test_blocks_that_differ_catches_a_shifted_offset():
    """bench.py's self-validation of the exchange (every rank, every block): the exact gather passes, a gather whose
    blocks land one column late (a wrong peer offset) or whose foreign block was never written fails - in the FOREIGN
    blocks, which the round-3 check (own block only) could not see."""
    from mxx_amd.parallel import blocks_that_differ

    world, rows, total = 4, 3, 10
    whole = np.arange(rows * total * 2).reshape(rows, total, 2)
    eq = lambda m: (lambda lo, hi: np.array_equal(m[:, lo:hi], whole[:, lo:hi]))
    assert blocks_that_differ(total, world, eq(whole.copy())) == []
    shifted = np.roll(whole, 1, axis=1)
    assert blocks_that_differ(total, world, eq(shifted)) == [0, 1, 2, 3]
    stale = whole.copy()
    sr = shard_range(total, world, 2)
    stale[:, sr.start:sr.stop] = 0  # rank 2's block never arrived
    assert blocks_that_differ(total, world, eq(stale)) == [2]
    # an own-block-only check on rank 0 would have passed both: its block is exact in `stale`
    s0 = shard_range(total, world, 0)
    assert np.array_equal(stale[:, s0.start:s0.stop], whole[:, s0.start:s0.stop])
    assert blocks_that_differ(7, 8, lambda lo, hi: True) == []  # an empty shard (7 columns on 8 ranks) is skipped


def test_world_size_2_gloo_foreign_block_validation(tmp_path):
    """two processes: each validates the WHOLE gathered product (its own block and the foreign one) against a
    recomputation, as bench.py does; a gather with rank 1's block misplaced is caught on rank 0"""
    script = tmp_path / "worker2.py"
    script.write_text(textwrap.dedent(
        """
        import sys
        sys.path.insert(0, {root!r})
        import numpy as np, torch, torch.distributed as dist
        from mxx_amd.parallel import shard_range, blocks_that_differ, place_column_blocks, padded_len
        from oracle import oracle as O
        dist.init_process_group(backend="gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        n, moduli = 16, O.gen_crt_basis(16, 2, 17)
        A = O.matrix_ntt(O.random_matrix(1, 2, 3, moduli, n), moduli)
        B = O.matrix_ntt(O.random_matrix(2, 3, 7, moduli, n), moduli)   # same seed on every rank: B is recomputable
        sr, pad = shard_range(7, world, rank), padded_len(7, world)
        block = np.zeros((2, pad, len(moduli), n), dtype=np.uint64)
        block[:, :len(sr)] = O.matmul(A, np.ascontiguousarray(B[:, sr.start:sr.stop]), moduli)
        recv = torch.empty(world * block.size, dtype=torch.int64)
        dist.all_gather_into_tensor(recv, torch.from_numpy(block.reshape(-1).view(np.int64)))
        g = recv.numpy().view(np.uint64).reshape((world,) + block.shape)
        whole = O.matmul(A, B, moduli)
        full = place_column_blocks(g, 7)
        eq = lambda m: (lambda lo, hi: np.array_equal(m[:, lo:hi], whole[:, lo:hi]))
        assert blocks_that_differ(7, world, eq(full)) == []
        wrong = full.copy()
        o = shard_range(7, world, 1)
        wrong[:, o.start:o.stop] = np.roll(full[:, o.start:o.stop], 1, axis=1)   # rank 1's block lands shifted
        assert blocks_that_differ(7, world, eq(wrong)) == [1]
        ok = torch.tensor([1.0]); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if rank == 0: print("FOREIGN_OK", float(ok.item()))
        dist.destroy_process_group()
        """).format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "FOREIGN_OK 1.0" in out.stdout


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun and no RANK/WORLD_SIZE in the environment must spawn two ranks
    that reach init_process_group (gloo + --dry-run here: no GPU in this container) and all-reduce."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--dist-backend", "gloo"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    got = json.loads(line)
    assert got == {"dry_run": True, "world": 2, "all_reduce": 2.0, "backend": "gloo"}
