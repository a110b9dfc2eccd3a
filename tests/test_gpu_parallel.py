"""GPU: zero-copy hand-off of engine memory to RCCL (world_size 1 here; the driver runs N=1..8).

Runs in a child process that imports torch BEFORE libgpupoly is loaded: the torch wheel
bundles its own libamdhip64.so.7 and a process must hold exactly one HIP runtime (bench.py
follows the same order for --gpus N>1)."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    import mxx_amd as mx
    from mxx_amd.parallel import DeviceBuffer
    from oracle import oracle as O
    n = 1024
    moduli = O.gen_crt_basis(n, 2, 24)
    p = mx.GpuDCRTPolyParams(n, moduli, 12)
    x = O.random_matrix(200, 2, 3, moduli, n)
    m = mx.GpuDCRTPolyMatrix.from_rns(p, x, False)
    m.ntt_all_in_place()
    mx.gpu_device_sync()
    t = DeviceBuffer(m).tensor(0)
    assert t.numel() == 2 * 3 * 2 * n * 4
    out = torch.empty_like(t)
    dist.all_gather_into_tensor(out, t)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32).reshape(2, 3, 2, n).astype(np.uint64)
    assert np.array_equal(got, O.matrix_ntt(x, moduli)), "RCCL saw different bytes than the engine wrote"
    dist.destroy_process_group()
    print("RCCL_ZERO_COPY_OK")
    """
)


def test_device_buffer_all_gather_world1(gpu, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "RCCL_ZERO_COPY_OK" in out.stdout


CHILD_GATHER = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    import mxx_amd as mx
    from mxx_amd.parallel import ColumnAllGather
    from oracle import oracle as O
    n = 4096
    moduli = O.gen_crt_basis(n, 3, 24)
    p = mx.GpuDCRTPolyParams(n, moduli, 12)
    for rows in (1, 3):   # 1 row: the block is received in place; 3 rows: staging + copy_block placement
        g = ColumnAllGather(p, rows, 5, 2, torch, dist, 0, slots=2)
        a = [mx.GpuDCRTPolyMatrix.from_rns(p, O.random_matrix(300 + i, rows, 4, moduli, n), True) for i in range(2)]
        b = [mx.GpuDCRTPolyMatrix.from_rns(p, O.random_matrix(310 + i, 4, 5, moduli, n), True) for i in range(6)]
        outs = [mx.GpuDCRTPolyMatrix(p, rows, 5, 2, True) for _ in range(2)]
        pending, fulls, want = {{}}, [], []
        # no host synchronisation anywhere in this loop: products, gathers and buffer re-use are ordered on the device
        for i in range(6):
            slot = i & 1
            if slot in pending:
                fulls.append(g.finish(pending.pop(slot)).clone())
            mx._ffi.check_status(mx._ffi.lib().gpu_matrix_mul(outs[slot].raw, a[slot].raw, b[i].raw), "gpu_matrix_mul")
            want.append(a[slot] * b[i])
            pending[slot] = g.start(outs[slot], slot)
        for slot in sorted(pending, key=lambda s: 4 + s):
            fulls.append(g.finish(pending[slot]).clone())
        assert len(fulls) == 6
        for i, (f, w) in enumerate(zip(fulls, want)):
            assert f == w, ("gathered product differs", rows, i)
        # a COEFF block: both sides of the ABI must carry the tag (ensure_eval would otherwise transform nothing)
        c = O.random_matrix(330 + rows, rows, 5, moduli, n)
        got = g.gather(mx.GpuDCRTPolyMatrix.from_rns(p, c, False))
        assert not got.is_ntt and got == mx.GpuDCRTPolyMatrix.from_rns(p, c, False), ("COEFF gather", rows)
        assert np.array_equal(got.ensure_eval().to_rns(), O.matrix_ntt(c, moduli)), ("COEFF gather, then NTT", rows)
    dist.destroy_process_group()
    print("RCCL_PIPELINED_GATHER_OK")
    """
)


def test_column_all_gather_is_ordered_on_the_device_world1(gpu, tmp_path):
    """ColumnAllGather.start / finish with two slots, no host synchronisation between the engine's kernels and the
    collective: every gathered matrix must equal the product that fed it (a missing stream dependency shows up as a
    stale or half-written block)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "child_gather.py"
    script.write_text(CHILD_GATHER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "RCCL_PIPELINED_GATHER_OK" in out.stdout


CHILD_UNEVEN = textwrap.dedent(
    """
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch
    torch.cuda.set_device(0)
    import mxx_amd as mx
    from mxx_amd.parallel import ColumnAllGather, shard_range
    from oracle import oracle as O

    class Work:
        def __init__(self, hub): self.hub = hub
        def wait(self): self.hub.deliver()

    class Hub:
        # two "ranks" of one process: the collective copies every rank's send buffer into every rank's receive buffer
        def __init__(self, world): self.world, self.calls = world, {{}}
        def register(self, rank, recv, send): self.calls[rank] = (recv, send)
        def deliver(self):
            if len(self.calls) < self.world: raise RuntimeError("a rank did not start its gather")
            torch.cuda.synchronize()
            n = self.calls[0][1].numel()
            for rank, (recv, _) in self.calls.items():
                for r in range(self.world):
                    recv[r * n:(r + 1) * n].copy_(self.calls[r][1])
            torch.cuda.synchronize()

    class FakeDist:
        def __init__(self, hub, rank): self.hub, self.rank = hub, rank
        def get_world_size(self): return self.hub.world
        def get_rank(self): return self.rank
        def all_gather_into_tensor(self, recv, send, async_op=False):
            self.hub.register(self.rank, recv, send)
            return Work(self.hub)

    n = 1024
    moduli = O.gen_crt_basis(n, 2, 24)
    p = mx.GpuDCRTPolyParams(n, moduli, 12)
    for rows, cols in ((1, 5), (3, 5), (22, 7)):   # 5 columns over 2 ranks: 3 + 2; 7: 4 + 3 (uneven shards, padded blocks)
        world = 2
        full = O.random_matrix(400 + rows, rows, cols, moduli, n)
        hub = Hub(world)
        gathers, pend = [], []
        for r in range(world):
            sr = shard_range(cols, world, r)
            local = mx.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(full[:, sr.start:sr.stop]), True)
            g = ColumnAllGather(p, rows, cols, 1, torch, FakeDist(hub, r), 0)
            gathers.append(g)
            pend.append(g.start(local))
        for r in range(world):
            got = gathers[r].finish(pend[r])
            assert np.array_equal(got.to_rns(), full), ("rank", r, "rows", rows, "cols", cols)
    print("UNEVEN_SHARDS_OK")
    """
)


def test_column_all_gather_uneven_shards_two_ranks_in_one_process(gpu, tmp_path):
    """The padded path of ColumnAllGather with uneven shards (what ranks see for 50 target columns on 8 GPUs): two
    ranks simulated in one process with a stand-in for torch.distributed that copies every rank's send buffer into
    every rank's receive buffer; the staging copy, the placement and the un-padding must rebuild the full matrix."""
    script = tmp_path / "child_uneven.py"
    script.write_text(CHILD_UNEVEN.format(root=ROOT))
    out = subprocess.run([sys.executable, str(script)], env=dict(os.environ), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "UNEVEN_SHARDS_OK" in out.stdout


def test_column_all_gather_two_processes_one_gpu_gloo(gpu):
    """Two ranks on the one GPU of the box (gloo on device tensors - RCCL refuses two ranks per device): the column
    gather with real inter-process data - in-place path, padded path, uneven shards (tools/rehearse_world2_gloo.py)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "tools", "rehearse_world2_gloo.py")],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert out.stdout.count("WORLD2_GLOO_GATHER_OK") == 2
