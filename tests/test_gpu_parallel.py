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
