"""Single-node multi-GPU sharding of the ring-matrix path (one process per GPU).

The reference scales by replicating read-only operands and sharding independent work
items through the host (`params_for_device`, `preimage_batched_sharded`,
src/sampler/trapdoor/gpu.rs:371-397).  Here the same partition is
expressed rank-wise for `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests):
  * target columns / output column blocks / polynomial batches are split into
    contiguous, balanced ranges — no data-path collective;
  * the one real exchange step, when a consumer needs the whole product on every GPU, is
    an all-gather of the output column blocks.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class ShardRange:
    start: int
    stop: int

    def __len__(self) -> int:
        return self.stop - self.start


def shard_range(total: int, world: int, rank: int) -> ShardRange:
    """Contiguous balanced split: the first `total % world` ranks get one extra unit
    (50 target columns on 8 GPUs -> 7,7,6,6,6,6,6,6)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return ShardRange(start, start + base + (1 if rank < extra else 0))


def all_shard_ranges(total: int, world: int) -> list[ShardRange]:
    return [shard_range(total, world, r) for r in range(world)]


def blocks_that_differ(cols_total: int, world: int, block_equal) -> list[int]:
    """Self-validation of a column all-gather: the ranks whose block of the gathered matrix differs from the
    expectation.  `block_equal(start, stop) -> bool` compares columns [start, stop) of the gathered matrix with a
    recomputation (or, for randomised results, checks the predicate the columns must satisfy); empty shards are skipped.
    bench.py runs it on EVERY rank over EVERY block, so a wrong peer offset, a missed wait or a stride bug in the
    exchange fails the run instead of leaving garbage in the foreign 7/8 of the matrix (`src/sampler/trapdoor/gpu.rs:371-397`
    is the fan-out whose results the exchange assembles)."""
    bad = []
    for r, sr in enumerate(all_shard_ranges(cols_total, world)):
        if len(sr) and not block_equal(sr.start, sr.stop):
            bad.append(r)
    return bad


def all_gather_column_blocks(local_block, cols_total: int, dist, device_tensor_of, world: int):
    """All-gather equally sized flattened column blocks.

    `local_block` is a 1-D torch tensor (bytes of this rank's rows x cols_local block,
    row-major polys); returns the list of per-rank 1-D tensors in rank order.  Callers
    with unequal shards pad to the largest shard (see `padded_len`)."""
    import torch

    out = torch.empty(world * local_block.numel(), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(out, local_block)
    return list(out.chunk(world))


def padded_len(total: int, world: int) -> int:
    return -(-total // world)


def place_column_blocks(gathered, cols_total: int):
    """Undo the padding of an all-gathered column-sharded matrix.

    `gathered` has shape (world, rows, padded_cols, ...) - rank r's rows x len(shard r) block sits in
    gathered[r, :, :len(shard r)], the rest of its columns is padding.  Returns the rows x cols_total
    matrix (numpy or torch, whatever came in).  The engine-side gather (ColumnAllGather) moves exactly
    these blocks with copy_block."""
    world = gathered.shape[0]
    parts = []
    for r, sr in enumerate(all_shard_ranges(cols_total, world)):
        if len(sr):
            parts.append(gathered[r][:, : len(sr)])
    if hasattr(gathered, "numpy") and not hasattr(gathered, "__array_interface__"):
        import torch

        return torch.cat(parts, dim=1)
    import numpy as np

    return np.concatenate(parts, axis=1)


class DeviceBuffer:
    """Zero-copy view of a libgpupoly matrix as a torch tensor (`__cuda_array_interface__`),
    so RCCL collectives run directly on the engine's HBM allocation.

    Ordering contract (the engine's stream is not torch's default stream): either issue the torch work under
    `torch.cuda.stream(engine_stream(...))`, which orders it against the engine's kernels on the device (what
    ColumnAllGather does), or synchronise on the host in BOTH directions - the engine's work that produced the
    bytes must have completed before a collective reads them (`gpu_device_sync()`), and the collective must have
    completed (e.g. `torch.cuda.current_stream().synchronize()`) before any engine call writes, frees or re-uses
    the viewed matrix.  The matrix is kept alive by this object; keep the object alive until then."""

    def __init__(self, matrix):
        import ctypes as C

        from . import _ffi

        ptr, size = C.c_void_p(), C.c_size_t()
        _ffi.check_status(_ffi.lib().gpupoly_matrix_device_ptr(matrix.raw, C.byref(ptr), C.byref(size)), "gpupoly_matrix_device_ptr")
        self._keep = matrix
        self.nbytes = size.value
        self.__cuda_array_interface__ = {
            "shape": (size.value,),
            "typestr": "|u1",
            "data": (ptr.value or 0, False),
            "version": 2,
        }

    def tensor(self, device_index: int):
        import torch

        return torch.as_tensor(self, device=torch.device("cuda", device_index))


def engine_stream(params, torch, device_index: int):
    """The engine context's compute stream as a torch stream: work issued under `torch.cuda.stream(...)` of it - a
    collective in particular - is ordered against the engine's kernels on the device, with no host synchronisation."""
    return torch.cuda.ExternalStream(params.ctx().stream_handle(), device=torch.device("cuda", device_index))


class ColumnAllGather:
    """All-gather of the column blocks of a rows x cols_total matrix that is sharded over the ranks by
    `shard_range` - the one exchange step of the sharded product / sharded preimage (SURVEY.md 8e;
    the reference moves these blocks through host bytes, src/sampler/trapdoor/gpu.rs:371-397).

    RCCL reads and writes the engine's own HBM allocations (DeviceBuffer).  With one row and equal
    shards every rank's block is a contiguous run of the full matrix, which is then the receive
    buffer itself; otherwise blocks are padded to the largest shard, gathered into a staging matrix
    and moved into place with copy_block.

    Ordering is on the device: the collective is issued with the engine's stream as torch's current stream, so it
    starts after the kernels that produced the block and `finish` makes the engine's stream wait for it - the host
    never blocks.  `start` / `finish` with `slots` > 1 let the gather of one step run under the compute of the next
    (xGMI transfer and kernels overlap); `gather` is the two back to back."""

    def __init__(self, params, rows: int, cols_total: int, level: int, torch, dist, device_index: int, slots: int = 1):
        from .matrix import GpuDCRTPolyMatrix

        self.torch, self.dist, self.device_index = torch, dist, device_index
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.rows, self.cols_total = rows, cols_total
        self.ranges = all_shard_ranges(cols_total, self.world)
        self.padded = padded_len(cols_total, self.world)
        self.direct = rows == 1 and all(len(r) == self.padded for r in self.ranges)
        self._stream = engine_stream(params, torch, device_index)
        self._slots = []
        for _ in range(max(1, slots)):
            slot = {"full": GpuDCRTPolyMatrix(params, rows, cols_total, level, True)}
            slot["full_t"] = DeviceBuffer(slot["full"]).tensor(device_index)
            if not self.direct:
                slot["send"] = GpuDCRTPolyMatrix(params, rows, self.padded, level, True)
                slot["recv"] = GpuDCRTPolyMatrix(params, self.world * rows, self.padded, level, True)
                slot["send_t"] = DeviceBuffer(slot["send"]).tensor(device_index)
                slot["recv_t"] = DeviceBuffer(slot["recv"]).tensor(device_index)
            self._slots.append(slot)

    @property
    def full(self):
        return self._slots[0]["full"]

    def start(self, local, slot: int = 0):
        """Enqueue the gather of `local` (this rank's rows x len(shard) block, EVAL or COEFF) into buffer `slot`.
        Returns a handle for `finish`; `local` and the slot's buffers must not be written until then."""
        s = self._slots[slot]
        mine = self.ranges[self.rank]
        assert local.nrow == self.rows and local.ncol == len(mine), "local block does not match this rank's shard"
        if self.direct:
            send_t, recv_t = DeviceBuffer(local).tensor(self.device_index), s["full_t"]
        else:
            send_t, recv_t = s["send_t"], s["recv_t"]
            if len(mine) == self.padded:
                send_t = DeviceBuffer(local).tensor(self.device_index)  # a full-width shard is its own send buffer
            elif len(mine):
                s["send"].copy_block_from(local, 0, 0, 0, 0, self.rows, len(mine))
        with self.torch.cuda.stream(self._stream):  # the collective waits for the engine's queued work
            work = self.dist.all_gather_into_tensor(recv_t, send_t, async_op=True)
        return (work, slot, local, send_t)

    def finish(self, pending):
        """The engine's stream waits (on the device) for the gather; returns the full matrix of its slot."""
        work, slot, local, _send_t = pending
        s = self._slots[slot]
        with self.torch.cuda.stream(self._stream):
            work.wait()
        if not self.direct:
            for r, sr in enumerate(self.ranges):
                if len(sr):
                    s["full"].copy_block_from(s["recv"], 0, sr.start, r * self.rows, 0, self.rows, len(sr))
        # RCCL moved raw bytes and copy_block propagated recv's tag: retag the C side from `local` (a zero-sized block copy
        # carries the source's format to the whole destination and moves nothing), then the Python mirror
        from . import _ffi

        _ffi.check_status(_ffi.lib().gpu_matrix_copy_block(s["full"].raw, local.raw, 0, 0, 0, 0, 0, 0), "gpu_matrix_copy_block")
        s["full"].is_ntt = local.is_ntt
        return s["full"]

    def gather(self, local, slot: int = 0):
        """local: this rank's rows x len(shard) block.  Returns the full matrix, valid on the engine's stream."""
        return self.finish(self.start(local, slot))


class GpuComm:
    """Communicator over the device contexts of THIS process (`gpupoly_comm_create`): the reference's own multi-GPU
    model - one process, a context per device (`params_for_device`, src/poly/dcrt/gpu.rs:531-557), rayon over them
    (src/sampler/trapdoor/gpu.rs:371-397) - with the exchange step on the devices (RCCL over xGMI, or event-ordered
    peer pulls when contexts share a device) instead of through host bytes.  No torch involved."""

    def __init__(self, params_list):
        import ctypes as C

        from . import _ffi

        self.params = list(params_list)
        arr = (C.c_void_p * len(self.params))(*[p.ctx_raw() for p in self.params])
        raw = C.c_void_p()
        _ffi.check_status(_ffi.lib().gpupoly_comm_create(arr, len(self.params), C.byref(raw)), "gpupoly_comm_create")
        self.raw = raw

    def __len__(self):
        return len(self.params)

    @property
    def backend(self) -> str:
        from . import _ffi

        return _ffi.lib().gpupoly_comm_backend(self.raw).decode()

    def all_gather_columns(self, local_blocks, fulls=None):
        """local_blocks[r]: rows x c_r block living in context r.  Returns the list of full matrices (one per context),
        valid on each context's stream; `fulls` re-uses caller-owned outputs."""
        import ctypes as C

        from . import _ffi
        from .matrix import GpuDCRTPolyMatrix

        n = len(self.params)
        assert len(local_blocks) == n, "one block per context"
        total = sum(b.ncol for b in local_blocks)
        if fulls is None:
            b0 = local_blocks[0]
            fulls = [GpuDCRTPolyMatrix(p, b0.nrow, total, b0.level, b0.is_ntt) for p in self.params]
        la = (C.c_void_p * n)(*[b.raw for b in local_blocks])
        fa = (C.c_void_p * n)(*[f.raw for f in fulls])
        _ffi.check_status(_ffi.lib().gpupoly_matrix_all_gather_columns(self.raw, la, fa), "gpupoly_matrix_all_gather_columns")
        for f in fulls:
            f.is_ntt = local_blocks[0].is_ntt
        return fulls

    def close(self):
        from . import _ffi

        if self.raw:
            _ffi.lib().gpupoly_comm_destroy(self.raw)
            self.raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
