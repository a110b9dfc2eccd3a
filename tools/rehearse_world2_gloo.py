"""Two ranks on ONE GPU (the box has one), gloo backend on device tensors: ColumnAllGather with real data across two
processes - the direct path (1 row, equal shards), the padded path (3 rows) and uneven shards (7 columns over 2 ranks).
Launch: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/rehearse_world2_gloo.py
RCCL refuses two ranks on one device, so this checks the partition / placement logic with real inter-process data, not RCCL."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group(backend="gloo")
import mxx_amd as mx
from mxx_amd.parallel import ColumnAllGather, shard_range
from oracle import oracle as O

rank, world = dist.get_rank(), dist.get_world_size()
n = 1024
moduli = O.gen_crt_basis(n, 3, 24)
p = mx.GpuDCRTPolyParams(n, moduli, 12)
ok = True
for rows, cols in ((1, 8), (3, 8), (1, 7), (2, 5)):
    full = O.random_matrix(500 + rows * 10 + cols, rows, cols, moduli, n)  # the same on every rank
    mine = shard_range(cols, world, rank)
    local = mx.GpuDCRTPolyMatrix.from_rns(p, np.ascontiguousarray(full[:, mine.start:mine.stop]), True)
    g = ColumnAllGather(p, rows, cols, 2, torch, dist, 0)
    got = g.gather(local)
    mx.gpu_device_sync()
    same = np.array_equal(got.to_rns(), full)
    print(f"rank {rank}: rows={rows} cols={cols} shard={mine.start}:{mine.stop} direct={g.direct} gathered==full: {same}", flush=True)
    ok = ok and same
dist.barrier()
dist.destroy_process_group()
if not ok:
    sys.exit(1)
print(f"rank {rank}: WORLD2_GLOO_GATHER_OK", flush=True)
