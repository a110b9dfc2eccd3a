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


class DeviceBuffer:
    """Zero-copy view of a libgpupoly matrix as a torch tensor (`__cuda_array_interface__`),
    so RCCL collectives run directly on the engine's HBM allocation."""

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
