#!/usr/bin/env python3
"""bench.py — DCRT poly-matrix mul ring-ops/s + trapdoor preimages/s on MI355X (BASELINE.json's metric),
next to the kernel's roofline and the CPU path timed on the same box.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload default|m1|m2a|m2b|m3a|m3b|m4]
                    [--scaling strong|weak] [--repeats R] [--no-cpu-baseline]

Default (what the driver runs): the two shapes the metric is quoted on, in ONE JSON line
  * main block  = M2A, the reference's benches/bench_matrix_mul_gpu.rs:20-34 shape: n=2^14, L=15 (24-bit),
    (1x30)*(30x120) in EVAL form = 3600 ring-ops per step, metric dcrt_ring_ops_per_s
    (1 ring-op = one R_q multiply-accumulate = n*L modular MACs);
  * "preimage"  = M3A, benches/bench_preimage_gpu.rs:7-56: n=2^14, L=10, base 2^12, sigma=4.578, d=1,
    50 uniform target columns per call, unit preimages/s (target columns per second);
  * "kernels"   = the NTT / INTT / pointwise mod-mul kernels of BASELINE configs[1] (M1: n=2^14, L=4,
    1024 polynomials) with their achieved HBM GB/s from hipEvents.
Always synchronises before stopping the clock (the reference's mat-mul bench does not).

Timing: W untimed warm-up steps, then EXACTLY K steps between barrier + device-synchronise brackets, MAX over
ranks (`value`, `ms_per_step`); after that R further repetitions of the K steps give `repeats` (median / min)
- BASELINE.md section 4 asks for both.  `roofline.kernel` is the kernel with the largest hipEvent total.

N>1: `python bench.py --gpus N` spawns its N ranks itself (before any GPU call; the driver's torchrun launch is
detected through WORLD_SIZE and used as is).  One process per GPU; `--scaling strong` (default) splits the
SAME problem: B/C column blocks (120 -> 15 per GPU at N=8) and target columns (50 -> 7,7,6,...) by
mxx_amd.parallel.shard_range; the blocks stay sharded between steps and one RCCL all-gather over xGMI per timed
region (inside it) assembles the whole matrix (`--gather lazy`, SURVEY.md 8e: the consumer gathers when it needs it;
`--gather step` exchanges after every step, overlapped with the next); `--scaling weak` runs the full shape on every
rank with no collective.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s
N_RING = 16384


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="default", choices=["default", "m1", "m2a", "m2b", "m2b_decompose", "m2b_mul_decompose", "m3a", "m3b", "m3a_refseq", "m4", "m4_batched"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--gather", default="step", choices=["lazy", "step"],
                    help="sharded runs: all-gather the column blocks (product, preimage) after every step, overlapped with the "
                         "next one (default: every step's result exists on every GPU), or once per timed region - the consumer "
                         "gathers when it needs the whole matrix; the line carries the other mode's figure as `other_gather`")
    ap.add_argument("--inproc", action="store_true",
                    help="N>1 in ONE process, as the reference runs (a context per device, a worker thread per context, the "
                         "exchange through gpupoly_matrix_all_gather_columns - RCCL behind the C ABI, no torch)")
    ap.add_argument("--repeats", type=int, default=10, help="extra repetitions of the K steps for median / min")
    ap.add_argument("--sustain", type=float, default=5.0,
                    help="seconds of one contiguous region of the headline workload after the timed K steps (0 = none)")
    ap.add_argument("--no-trace", action="store_true", help="skip the per-kernel launch trace (composed rooflines)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget per block")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo rehearses the launcher / rendezvous on a GPU-less box (with --dry-run) and the whole sharded path with more ranks than GPUs (ranks share devices; not a measurement)")
    ap.add_argument("--dry-run", action="store_true", help="launch + rendezvous + one all-reduce, no GPU work")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------
# launcher: --gpus N without torchrun
# ---------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args) -> int:
    """Spawn one fresh child per rank (this process never touches the GPU), relay rank 0's output,
    fail if any rank fails."""
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT") or _free_port())
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        out = None if rank == 0 else subprocess.DEVNULL  # rank 0 prints the JSON line; stderr stays visible
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    except KeyboardInterrupt:
        rc = 130
    finally:
        for p in procs:  # only the children started here, by PID
            if p.poll() is None:
                p.terminate()
    return rc


class InprocRank:
    """One of N device contexts driven by ONE process (--inproc): the reference's own model (`params_for_device`,
    src/poly/dcrt/gpu.rs:531-557; rayon over the contexts, src/sampler/trapdoor/gpu.rs:371-397).  The exchange step goes
    through gpupoly_matrix_all_gather_columns (RCCL behind the C ABI); no torch in the process."""

    inproc, active, torch, dist, backend = True, True, None, None, "inproc"

    def __init__(self, world, rank, dnum=None):
        self.world, self.rank, self.local_rank, self.dnum = world, rank, rank, dnum


class Dist:
    """torch.distributed when the job has more than one rank (RCCL = backend "nccl" on ROCm)."""

    inproc, dnum = False, None

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = self.dist = None
        self.backend = args.dist_backend
        if self.world > 1 or os.environ.get("MXX_BENCH_FORCE_DIST") == "1":  # the env rehearses the N>1 path on one GPU
            # torch must be imported BEFORE libgpupoly is loaded: the wheel bundles its own libamdhip64 and
            # a process must hold exactly one HIP runtime (libgpupoly then binds to the copy torch mapped)
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(backend="gloo")
            self.torch, self.dist = torch, dist
            self.world = dist.get_world_size()
            self.rank = dist.get_rank()
            # the first collective of a communicator sets up its channels (tens to hundreds of milliseconds): run one of
            # each kind used below now, so that none of that can land in a timed region even with --warmup 0
            dev = self.device()
            one = torch.ones(1024, dtype=torch.int32, device=dev)
            out = torch.empty(1024 * self.world, dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(out, one)
            dist.all_reduce(torch.ones(1, dtype=torch.float64, device=dev), op=dist.ReduceOp.MAX)
            dist.barrier()
            if self.backend == "nccl":
                torch.cuda.synchronize()

    @property
    def active(self):
        return self.torch is not None

    def device(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def barrier_sync(self, sync_engine):
        sync_engine()
        if self.active:
            if self.backend == "nccl":
                self.torch.cuda.synchronize()
            self.dist.barrier()
            if self.backend == "nccl":
                self.torch.cuda.synchronize()

    def max_over_ranks(self, value: float) -> float:
        if not self.active:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, value: float) -> float:
        if not self.active:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def gather_objects(self, obj):
        if not self.active:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def finish(self):
        if self.active:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
def fixed_seed(mx, tag: int):
    return mx.GpuRngSeed.from_bytes(bytes([(tag * 37 + i * 11 + 5) & 0xFF for i in range(32)]))


def uniform_matrix(mx, params, rows, cols, tag, total_cols=None, col_start=0):
    """Synthetic i.i.d. uniform residues, generated on the device (EVAL form), deterministic in `tag`;
    a column window of the same logical matrix when total_cols is given."""
    code = mx.DistType.FinRingDist().as_ffi()
    if total_cols is None:
        return mx.GpuDCRTPolyMatrix.sample_distribution(params, rows, cols, code, 0.0, fixed_seed(mx, tag))
    return mx.GpuDCRTPolyMatrix.sample_distribution_columns(params, rows, total_cols, col_start, cols, code, 0.0,
                                                             fixed_seed(mx, tag))


def inject_fault(full):
    """Test hook (tests/test_gpu_comm.py): MXX_BENCH_FAULT_INJECT=shift hands the self-validation a gathered matrix whose
    columns are rotated by one - what a wrong peer offset in the exchange would produce - so that the suite can prove the
    bench fails on it.  Unset (always, outside that test): the matrix itself."""
    if os.environ.get("MXX_BENCH_FAULT_INJECT") != "shift" or full.ncol < 2:
        return full
    return full.slice_columns(1, full.ncol).concat_columns([full.slice_columns(0, 1)])


class Workload:
    """A step = one pass of the hot path over one batch; marks bracket its dominant kernel(s)."""

    metric = "dcrt_ring_ops_per_s"
    unit = "ring-ops/s"
    name = ""
    depth = 0
    kernels = ()  # ((label, algorithmic bytes per launch), ...) in the order step() marks them

    def __init__(self, mx, d: Dist, args, device: int):
        self.mx, self.d, self.args = mx, d, args
        # MXX_BENCH_FORCE_DIST=1 rehearses the sharded path (partition + RCCL gather) with a world of one rank
        self.strong = args.scaling == "strong" and (d.world > 1 or d.active)
        self.params = mx.GpuDCRTPolyParams(N_RING if not self.name.startswith("m4") else 256, self.moduli(mx), self.base_bits(),
                                           gpu_ids=[device], dnum=d.dnum)
        self.lazy_gather = args.gather == "lazy"
        self.ctx = self.params.ctx()
        self.word = self.ctx.word_bytes()
        self.device = device
        self.nmarks = len(self.kernels) + 1
        self._pending = {}
        self.full = None

    def moduli(self, mx):
        return mx.gen_crt_basis(N_RING, self.depth, 24)

    def base_bits(self):
        return 12

    def mark(self, step, j):
        self.ctx.timer_mark(step * self.nmarks + j)

    def kernel_ms(self, steps):
        """mean hipEvent duration of every marked kernel over the timed steps"""
        out = []
        for j in range(len(self.kernels)):
            out.append(statistics.fmean(self.ctx.timer_elapsed(s * self.nmarks + j, s * self.nmarks + j + 1) for s in range(steps)))
        return out

    def check(self):
        pass

    # -- sharded runs: the gather of step i runs under the compute of step i+1 (two buffer slots) ----------------
    gather = None

    def gather_begin(self, i):
        """before step i writes buffer slot i % 2: the engine's stream waits for the gather that last used it"""
        slot = i & 1
        if self.gather is not None and self._pending.get(slot) is not None:
            self.full = self.gather.finish(self._pending.pop(slot)[1])
        return slot

    def gather_enqueue(self, i, slot, local):
        if self.gather is not None:
            self._pending[slot] = (i, self.gather.start(local, slot))

    def drain(self):
        """finish every gather still in flight, oldest first (called before the clock stops)"""
        if self.gather is not None:
            for slot in sorted(self._pending, key=lambda s_: self._pending[s_][0]):
                self.full = self.gather.finish(self._pending[slot][1])
            self._pending.clear()


class M1(Workload):
    name, depth = "m1", 4

    def setup(self):
        mx, p = self.mx, self.params
        self.batch = 1024
        self.x = uniform_matrix(mx, p, self.batch, 1, 2)
        self.x.intt_all_in_place()
        self.w = uniform_matrix(mx, p, 1, 1, 3)
        self.units = self.batch * self.d.world  # weak only: batches are independent polynomials
        vec = 2.0 * N_RING * self.word * self.batch * self.depth  # 2*n*w per (poly, limb)
        self.vec_bytes = vec
        self.kernels = (("ntt14::fwd_kernel<u32> (forward negacyclic NTT, 2^14 points)", vec),
                        ("ntt14::inv_kernel<u32,signed,mulw> (pointwise product by a resident ring element + inverse NTT, fused)", vec))
        self.nmarks = 3
        self.desc = (f"M1 (BASELINE configs[1]): n=2^14, L=4 (24-bit), batch {self.batch} polys; step = x<-INTT(NTT(x) o w) = "
                     f"{self.batch} ring mults; two kernels (the product rides in the inverse transform's load)")
        self.sharding = "independent polynomial batches per rank, no collective"

    def step(self, i, mark):
        from mxx_amd import _ffi

        lib = _ffi.lib()
        if mark:
            self.mark(i, 0)
        _ffi.check_status(lib.gpu_matrix_ntt_all(self.x.raw), "gpu_matrix_ntt_all")
        self.x.is_ntt = True
        if mark:
            self.mark(i, 1)
        _ffi.check_status(lib.gpupoly_matrix_mul_scalar_intt(self.x.raw, self.x.raw, self.w.raw), "gpupoly_matrix_mul_scalar_intt")
        self.x.is_ntt = False
        if mark:
            self.mark(i, 2)

    def standalone_kernels(self, reps=10):
        """the unfused kernels of the same chain, each between its own hipEvent marks: point-wise mod-mul and the
        plain inverse transform (BASELINE.json asks for the NTT and mod-mul kernels' GB/s)"""
        from mxx_amd import _ffi

        lib, base = _ffi.lib(), 60000
        y = self.x.clone()
        _ffi.check_status(lib.gpu_matrix_ntt_all(y.raw), "gpu_matrix_ntt_all")
        # two passes: the first creates the hipEvents (creation on the host between two records would be timed)
        for _ in range(2):
            for r in range(reps):  # y <- y o w, repeatedly
                self.ctx.timer_mark(base + 2 * r)
                _ffi.check_status(lib.gpu_matrix_mul_scalar(y.raw, y.raw, self.w.raw), "gpu_matrix_mul_scalar")
                self.ctx.timer_mark(base + 2 * r + 1)
            mul_ms = [self.ctx.timer_elapsed(base + 2 * r, base + 2 * r + 1) for r in range(reps)]
        base += 2 * reps
        for _ in range(2):
            for r in range(reps):
                self.ctx.timer_mark(base + 2 * r)
                _ffi.check_status(lib.gpu_matrix_intt_all(y.raw), "gpu_matrix_intt_all")
                self.ctx.timer_mark(base + 2 * r + 1)
                _ffi.check_status(lib.gpu_matrix_ntt_all(y.raw), "gpu_matrix_ntt_all")
            inv_ms = [self.ctx.timer_elapsed(base + 2 * r, base + 2 * r + 1) for r in range(reps)]
        return statistics.fmean(mul_ms), statistics.fmean(inv_ms)


def large_batch_kernels(mx, device, polys=4096, reps=6):
    """The M1 kernels on a batch the Infinity Cache cannot hold (4096 polys x 4 limbs x 2^14 x 4 B = 1.07 GB; M1's own
    268 MB batch about equals the 256 MiB cache, which flatters its GB/s): forward / inverse transform, the fused
    product + inverse and the stand-alone mod-mul, hipEvent time per launch."""
    from mxx_amd import _ffi

    lib = _ffi.lib()
    p = mx.GpuDCRTPolyParams(N_RING, mx.gen_crt_basis(N_RING, 4, 24), 12, gpu_ids=[device])
    ctx = p.ctx()
    x = uniform_matrix(mx, p, polys, 1, 21)
    w = uniform_matrix(mx, p, 1, 1, 22)
    algo = 2.0 * N_RING * 4 * polys * 4
    calls = {
        "ntt_inverse": lambda: _ffi.check_status(lib.gpu_matrix_intt_all(x.raw), "intt"),
        "ntt_forward": lambda: _ffi.check_status(lib.gpu_matrix_ntt_all(x.raw), "ntt"),
        "mod_mul": lambda: _ffi.check_status(lib.gpu_matrix_mul_scalar(x.raw, x.raw, w.raw), "mul_scalar"),
        "mul_intt_fused": lambda: _ffi.check_status(lib.gpupoly_matrix_mul_scalar_intt(x.raw, x.raw, w.raw), "mul_scalar_intt"),
    }
    order = ["ntt_inverse", "ntt_forward", "mod_mul", "mul_intt_fused"]  # formats alternate EVAL -> COEFF -> EVAL -> EVAL -> COEFF
    base, ms = 50000, {k: [] for k in order}
    for rep in range(reps + 1):  # the first repetition creates the hipEvents and warms the kernels
        for j, k in enumerate(order):
            ctx.timer_mark(base + 2 * j)
            calls[k]()
            ctx.timer_mark(base + 2 * j + 1)
        if rep:
            for j, k in enumerate(order):
                ms[k].append(ctx.timer_elapsed(base + 2 * j, base + 2 * j + 1))
        _ffi.check_status(lib.gpu_matrix_ntt_all(x.raw), "ntt")  # back to EVAL for the next repetition
    out = {"batch": f"{polys} polys x 4 limbs x 2^14 (u32): {algo / 2e9:.2f} GB per operand, beyond the 256 MiB Infinity Cache"}
    for k in order:
        t = statistics.median(ms[k])
        gbs = algo / (t * 1e-3) / 1e9
        out[k] = {"us": round(t * 1e3, 1), "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                  "ns_per_vector": round(t * 1e6 / (polys * 4), 2)}
    return out


class MatMul(Workload):
    def setup(self):
        mx, p, d = self.mx, self.params, self.d
        r, k, c = self.shape
        L = self.depth
        self.a = uniform_matrix(mx, p, r, k, 4)
        if self.strong:
            from mxx_amd.parallel import ColumnAllGather, shard_range

            sr = shard_range(c, d.world, d.rank)
            self.c_local = len(sr)
            self.b = uniform_matrix(mx, p, k, self.c_local, 5, total_cols=c, col_start=sr.start)
            # --inproc: the runner gathers through gpupoly_matrix_all_gather_columns once every rank has enqueued its step
            self.gather = None if d.inproc else ColumnAllGather(p, r, c, L - 1, d.torch, d.dist, self.device, slots=2)
            how = ("the blocks stay sharded between steps and are all-gathered (RCCL, on the device's stream order) ONCE "
                   "per timed region, inside it - the full C is larger than what a rank reads for its product and xGMI is "
                   "~20x slower than HBM, so a consumer gathers when it needs the whole matrix (SURVEY 8e)"
                   if self.lazy_gather else
                   "one RCCL all-gather of C's blocks per step, ordered on the device and overlapped with the next "
                   "step's product (two buffer slots)")
            self.sharding = (f"strong: B/C column blocks by shard_range ({c} -> {self.c_local} on this rank), A replicated; " + how)
            self.units_total = r * k * c  # the whole job's ring-ops per step
        else:
            self.c_local = c
            self.b = uniform_matrix(mx, p, k, c, 5)
            self.gather = None
            self.sharding = "weak: every rank multiplies the full shape, no collective" if d.world > 1 else "single GPU"
            self.units_total = r * k * c * d.world
        self._ungathered = False
        self.outs = [mx.GpuDCRTPolyMatrix(p, r, max(self.c_local, 1), L - 1, True) for _ in range(2 if self.strong else 1)]
        self.out = self.outs[0]
        self.units = self.units_total
        algo = float(r * k + k * self.c_local + r * self.c_local) * N_RING * L * self.word  # SURVEY 8d, this rank's launch
        self.kernels = ((self.kernel_label, algo),)
        self.nmarks = 2
        self.desc = (f"{self.name.upper()}: n=2^14, L={L} (24-bit), ({r}x{k})*({k}x{c}) in EVAL form; "
                     "1 ring-op = one R_q multiply-accumulate")

    def step(self, i, mark):
        from mxx_amd import _ffi

        per_step = self.strong and not self.lazy_gather  # two output buffers: the gather of step i runs under step i + 1
        slot = (self.gather_begin(i) if self.gather is not None else i & 1) if per_step else 0
        self.out = self.outs[slot]
        if mark:
            self.mark(i, 0)
        if self.c_local:
            _ffi.check_status(_ffi.lib().gpu_matrix_mul(self.out.raw, self.a.raw, self.b.raw), "gpu_matrix_mul")
        if mark:
            self.mark(i, 1)
            if i == 0:  # the roofline names what the dispatcher launched for THIS product, not a constant
                self.launched_kernel = self.ctx.last_kernel()
        if per_step and self.gather is not None:
            self.gather_enqueue(i, slot, self.local_block())
        self._ungathered = True

    def local_block(self):
        return self.out if self.c_local else self.out.slice_columns(0, 0)

    def drain(self):
        if self.gather is not None and self.lazy_gather:
            if self._ungathered:  # the region's one exchange step
                self.full = self.gather.gather(self.local_block())
                self._ungathered = False
            return
        super().drain()

    def check(self):
        """size-independent property on the timed operands: (A*B) == columns of A*[B] recomputed entry-wise
        for one column through the ring (a second, different kernel path is used by the 1-column product)"""
        col = self.b.slice_columns(0, 1) if self.c_local else None
        if col is not None:
            one = self.a * col
            assert one == self.out.slice_columns(0, 1), "product column differs between kernel paths"
        if self.strong and self.full is not None:
            # self-validation of the exchange: EVERY rank recomputes the WHOLE product - B is a pure
            # function of (seed, global column), so the foreign blocks are recomputable here - and compares it with the
            # gathered matrix: a wrong peer offset, a missed event wait or a stride bug in the gather cannot pass
            from mxx_amd.parallel import blocks_that_differ

            r, k, c = self.shape
            assert self.full.ncol == c, "gathered product has the wrong width"
            whole = self.a * uniform_matrix(self.mx, self.params, k, c, 5)
            full = inject_fault(self.full)
            bad = blocks_that_differ(c, self.d.world, lambda lo, hi: full.slice_columns(lo, hi) == whole.slice_columns(lo, hi))
            if bad:
                raise AssertionError(f"rank {self.d.rank}: gathered product differs from the recomputed one in the blocks of ranks {bad}")
            self.foreign_blocks_checked = self.d.world - 1



class M2A(MatMul):
    name, depth, shape = "m2a", 15, (1, 30, 120)
    kernel_label = "R_q matrix product (skinny: B streamed once)"  # replaced by gpupoly_context_last_kernel() after the first step


class M2B(MatMul):
    name, depth, shape = "m2b", 8, (64, 64, 64)
    kernel_label = "R_q matrix product (fat)"


class M2BDecompose(Workload):
    """BASELINE configs[2], second half: G^-1 of a 64 x 64 matrix over R_q (n=2^14, L=8, base 2^12 -> k = 16 digits per
    entry): EVAL in, the 1024 x 64 digit matrix in EVAL form out (34.4 GB, written once).  The reference wrapper's
    `decompose` (src/matrix/gpu_dcrt_poly.rs:315-341: clone, INTT, gpu_matrix_decompose_base into an EVAL output)."""

    name, depth = "m2b_decompose", 8

    def setup(self):
        mx, p = self.mx, self.params
        self.m = uniform_matrix(mx, p, 64, 64, 6)
        self.k = p.modulus_digits()
        self.units = 64 * 64 * self.d.world  # ring elements decomposed per step (independent per rank)
        # (r c + r k c) n L w - the source read once, the digit matrix written once
        self.algo = float(64 * 64 + 64 * self.k * 64) * N_RING * self.depth * self.word
        self.kernels = (("decompose call (copy + INTT of the source, digits inside the forward transform's load)", self.algo),)
        self.nmarks = 2
        self.metric, self.unit = "ring_elements_decomposed_per_s", "ring-elements/s"
        self.desc = (f"M2B decompose (BASELINE configs[2]): n=2^14, L={self.depth} (24-bit), base 2^12, G^-1 of a 64x64 matrix -> "
                     f"{64 * self.k}x64 digit polynomials in EVAL form")
        self.sharding = "independent matrices per rank, no collective" if self.d.world > 1 else "single GPU"
        self.out = None

    def step(self, i, mark):
        if mark:
            self.mark(i, 0)
        self.out = None  # the previous step's 34 GB go back to the stream-ordered cache before the next allocation
        self.out = self.m.decompose()
        if mark:
            self.mark(i, 1)

    def check(self):
        g = self.mx.GpuDCRTPolyMatrix.gadget_matrix(self.params, 64)
        assert g * self.out == self.m, "G * G^-1(M) != M"


class M2BMulDecompose(Workload):
    """The fused consumer of the same decomposition: S * G^-1(B), S 8 x 1024, B 64 x 64 - one ABI
    call (gpupoly_matrix_mul_decompose) where the reference's wrapper loops over columns (gpu_dcrt_poly.rs:1414-1493)."""

    name, depth = "m2b_mul_decompose", 8

    def setup(self):
        mx, p = self.mx, self.params
        self.k = p.modulus_digits()
        self.b = uniform_matrix(mx, p, 64, 64, 6)
        self.s = uniform_matrix(mx, p, 8, 64 * self.k, 7)
        self.units = 8 * 64 * self.k * 64 * self.d.world  # ring multiply-accumulates per step
        # operands and result only: the digit matrix is internal to the call (its 34.4 GB are what a fusion that never
        # materialised it would save; the composed roofline below prices the kernels that actually run)
        self.algo = float(8 * 64 * self.k + 64 * 64 + 8 * 64) * N_RING * self.depth * self.word
        self.kernels = (("mul_decompose call (digit transform + product)", self.algo),)
        self.nmarks = 2
        self.desc = (f"M2B mul_decompose: n=2^14, L={self.depth}, (8x{64 * self.k}) * G^-1(64x64), one ABI call; "
                     "1 ring-op = one R_q multiply-accumulate")
        self.sharding = "independent products per rank, no collective" if self.d.world > 1 else "single GPU"
        self.out = None

    def step(self, i, mark):
        if mark:
            self.mark(i, 0)
        self.out = None
        self.out = self.s.mul_decompose(self.b)
        if mark:
            self.mark(i, 1)

    def check(self):
        assert self.out == self.s * self.b.decompose(), "S * G^-1(B) differs from the two-step form"


class Preimage(Workload):
    metric, unit = "trapdoor_preimages_per_s", "preimages/s"
    sigma, dsize, cols = 4.578, 1, 50

    def setup(self):
        mx, p, d = self.mx, self.params, self.d
        self.sampler = mx.GpuDCRTPolyTrapdoorSampler(p, self.sigma)
        from mxx_amd.sampler import seed_source

        # fixed seeds for the trapdoor so that every rank holds the SAME trapdoor / public matrix (replicated
        # read-only operands); the per-call sampler seeds stay OS-random as in the reference
        with seed_source(bytes([(7 * j + i) & 0xFF for i in range(32)]) for j in range(3)):
            self.td, self.pub = self.sampler.trapdoor(p, self.dsize)
        k = p.modulus_digits()
        if self.strong:
            from mxx_amd.parallel import ColumnAllGather, shard_range

            sr = shard_range(self.cols, d.world, d.rank)
            self.c_local = len(sr)
            self.target = uniform_matrix(mx, p, self.dsize, self.c_local, 9, total_cols=self.cols, col_start=sr.start)
            self.out_rows = (k + 2) * self.dsize
            self.gather = None if d.inproc else ColumnAllGather(p, self.out_rows, self.cols, self.depth - 1, d.torch, d.dist, self.device, slots=2)
            how = ("the preimage blocks stay on the device that sampled them between calls (the reference's fan-out, "
                   "src/sampler/trapdoor/gpu.rs:371-397, never moves them device to device) and are all-gathered (RCCL, on the "
                   "device's stream order) ONCE per timed region, inside it - a rank's block is 98 MB per call at N=8, an "
                   "exchange per call would be xGMI-bound" if self.lazy_gather else
                   "one RCCL all-gather of the preimage blocks per call, ordered on the device and overlapped with the next call")
            self.sharding = (f"strong: {self.cols} target columns by shard_range ({self.c_local} on this rank), trapdoor "
                             f"replicated; {how}")
            self.units = self.cols
        else:
            self.c_local = self.cols
            self.target = uniform_matrix(mx, p, self.dsize, self.cols, 9)
            self.gather = None
            self.sharding = "weak: every rank samples all 50 columns, no collective" if d.world > 1 else "single GPU"
            self.units = self.cols * d.world
        self.kernels = (("preimage call (all kernels; per-kernel split in profiles/)", None),)
        self.nmarks = 2
        self.desc = (f"{self.name.upper()}: bench_preimage shape n=2^14, L={self.depth}, base 2^12, sigma={self.sigma}, "
                     f"d=1, {self.cols} target columns per call")
        self.x = None

    def step(self, i, mark):
        per_call = self.gather is not None and not self.lazy_gather
        slot = self.gather_begin(i) if per_call else 0
        if mark:
            self.mark(i, 0)
        if self.c_local:
            self.x = self.sampler.preimage(self.params, self.td, self.pub, self.target)
        if mark:
            self.mark(i, 1)
        if self.strong:
            if not self.c_local:
                self.x = self.mx.GpuDCRTPolyMatrix(self.params, self.out_rows, 0, self.depth - 1, True)
            if per_call:
                self.gather_enqueue(i, slot, self.x)
            else:
                self._ungathered = True

    def local_block(self):
        return self.x

    def drain(self):
        if self.gather is not None and self.lazy_gather:
            if getattr(self, "_ungathered", False):  # the region's one exchange step
                self.full = self.gather.gather(self.x)
                self._ungathered = False
            return
        super().drain()

    def check(self):
        if self.c_local:
            assert self.pub * self.x == self.target, "A*x != u"
        if self.strong and self.full is not None:
            # self-validation of the exchange: the preimages are freshly randomised per call, so a foreign block cannot be
            # recomputed - but the trapdoor is replicated and the target is a pure function of (seed, global column), so
            # EVERY rank checks A * x = u over ALL gathered columns, and its own block bit for bit
            from mxx_amd.parallel import blocks_that_differ, shard_range

            assert self.full.ncol == self.cols, "gathered preimage has the wrong width"
            whole_target = uniform_matrix(self.mx, self.params, self.dsize, self.cols, 9)
            full = inject_fault(self.full)
            img = self.pub * full
            bad = blocks_that_differ(self.cols, self.d.world, lambda lo, hi: img.slice_columns(lo, hi) == whole_target.slice_columns(lo, hi))
            if bad:
                raise AssertionError(f"rank {self.d.rank}: A * x != u in the gathered blocks of ranks {bad}")
            sr = shard_range(self.cols, self.d.world, self.d.rank)
            if self.c_local:
                assert full.slice_columns(sr.start, sr.stop) == self.x, "gathered preimage block differs from the local one"
            self.foreign_blocks_checked = self.d.world - 1


class M3A(Preimage):
    name, depth = "m3a", 10


class M3B(Preimage):
    name, depth = "m3b", 8


class M3ARefSeq(M3A):
    """M3A through ONLY the entry points the reference's Rust side binds, in the reference's own order
    (src/sampler/trapdoor/gpu.rs:228-369, :423-474): what an unpatched mxx gets from this library.  The headline M3A block
    uses the extension sequence (row views, fused NTT + add, one stacked product)."""

    name = "m3a_refseq"

    def setup(self):
        super().setup()
        self.desc += "; the reference's own call sequence (no gpupoly_* extension entry)"

    def step(self, i, mark):
        if mark:
            self.mark(i, 0)
        self.x = self.sampler.preimage_reference_sequence(self.params, self.td, self.pub, self.target)
        if mark:
            self.mark(i, 1)


class M4(Workload):
    """BASELINE configs[4] flavour: the parameter family of tests/test_gpu_ggh15_modp_chain.rs:36-42 (n=256, 51-bit
    limbs, base 2^17, depth <= 12) as the chain those schemes run per level: preimage of a 2d-column target,
    encoding * key, mul_decompose - launch-bound (u64 words, ~100 small launches per step)."""

    name, depth = "m4", 12
    metric, unit = "trapdoor_preimages_per_s", "preimages/s"

    def moduli(self, mx):
        return mx.gen_crt_basis(256, self.depth, 51)

    def base_bits(self):
        return 17

    def setup(self):
        mx, p = self.mx, self.params
        d = 2
        self.dd = d
        self.sampler = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
        self.td0, self.a0 = self.sampler.trapdoor(p, d)
        _, a1 = self.sampler.trapdoor(p, d)
        self.target = a1.slice(0, d, 0, 2 * d)
        k = p.modulus_digits()
        us = mx.GpuDCRTPolyUniformSampler()
        self.c0 = us.sample_uniform(p, 1, self.a0.col_size(), mx.DistType.FinRingDist())
        self.bmat = us.sample_uniform(p, d, d * k, mx.DistType.FinRingDist())
        self.mmat = us.sample_uniform(p, d, 3, mx.DistType.FinRingDist())
        self.units = 2 * d * self.d.world  # target columns per step
        self.kernels = (("chain step (launch-bound; launches per step in config)", None),)
        self.nmarks = 2
        self.desc = (f"M4 (BASELINE configs[4] parameters): n=256, L={self.depth} (51-bit, u64 words), base 2^17, d={d}; step = "
                     f"preimage of {2 * d} columns + (1x{self.a0.col_size()})*K + mul_decompose({d}x{d * k}, {d}x3)")
        self.sharding = "independent chains per rank, no collective"

    def step(self, i, mark):
        from mxx_amd import _ffi

        if mark:
            self.mark(i, 0)
        n0 = _ffi.lib().gpupoly_launch_count()
        self.k = self.sampler.preimage(self.params, self.td0, self.a0, self.target)
        self.c1 = self.c0 * self.k
        self.md = self.bmat.mul_decompose(self.mmat)
        self.launches_per_step = _ffi.lib().gpupoly_launch_count() - n0
        if mark:
            self.mark(i, 1)

    def check(self):
        assert self.a0 * self.k == self.target, "A*x != u"
        assert self.mx.GpuDCRTPolyMatrix.gadget_matrix(self.params, self.dd) * self.mmat.decompose() == self.mmat


class M4Batched(M4):
    """The same chain with the requests a GGH15 caller actually holds at once (src/lookup/ggh15/pubkey_gpu.rs:615-971 hands
    `preimage_batched_sharded` dozens of targets per key): `requests` independent 4-column targets against one trapdoor per
    step, through `preimage_many` (one sequence of launches over the concatenated targets, every request's output identical
    to what it gets alone), then the requests' encoding products as one batched launch and their mul_decompose gates as one
    call over the concatenated operands (G^-1 works column by column)."""

    name = "m4_batched"
    requests = int(os.environ.get("MXX_BENCH_M4_REQUESTS", "16"))  # the default line reports 16 in flight

    def setup(self):
        super().setup()
        mx, p = self.mx, self.params
        us = mx.GpuDCRTPolyUniformSampler()
        self.targets = [self.target] + [us.sample_uniform(p, self.dd, 2 * self.dd, mx.DistType.FinRingDist()) for _ in range(self.requests - 1)]
        self.mmats = [self.mmat] + [us.sample_uniform(p, self.dd, 3, mx.DistType.FinRingDist()) for _ in range(self.requests - 1)]
        self.units = 2 * self.dd * self.requests * self.d.world
        self.desc = (f"M4 batched: {self.requests} requests per step, each = preimage of {2 * self.dd} columns + (1x{self.a0.col_size()})*K "
                     f"+ mul_decompose; n=256, L={self.depth} (51-bit, u64 words), base 2^17, d={self.dd}; preimage_many + mul_batch + one mul_decompose over the concatenated operands")

    def step(self, i, mark):
        from mxx_amd import _ffi

        if mark:
            self.mark(i, 0)
        n0 = _ffi.lib().gpupoly_launch_count()
        M = self.mx.GpuDCRTPolyMatrix
        self.ks = self.sampler.preimage_many(self.params, self.td0, self.a0, self.targets)
        self.c1s = M.mul_batch([self.c0] * self.requests, self.ks)  # the requests' encodings times their keys: one launch
        # B * G^-1(M_j) for every request: the gadget decomposition is column-wise, so the requests' M_j ride in one call
        self.mds = self.bmat.mul_decompose(M.concat_columns_of(self.mmats)).split_columns([3] * self.requests)
        self.k, self.md = self.ks[0], self.mds[0]
        self.launches_per_step = _ffi.lib().gpupoly_launch_count() - n0
        if mark:
            self.mark(i, 1)

    def check(self):
        for k_, t in zip(self.ks, self.targets):
            assert self.a0 * k_ == t, "A*x != u"
        assert self.c1s[3] == self.c0 * self.ks[3] and self.mds[5] == self.bmat.mul_decompose(self.mmats[5])


WORKLOADS = {"m1": M1, "m2a": M2A, "m2b": M2B, "m2b_decompose": M2BDecompose, "m2b_mul_decompose": M2BMulDecompose,
             "m3a": M3A, "m3b": M3B, "m3a_refseq": M3ARefSeq, "m4": M4, "m4_batched": M4Batched}


# ---------------------------------------------------------------------------------------------------
# measurement
# ---------------------------------------------------------------------------------------------------
def run_block(wl: Workload, d: Dist, steps: int, warmup: int, repeats: int, sustain_s: float = 0.0):
    """The contract's timed region (K steps, barrier + synchronise on both sides, MAX over ranks), then
    `repeats` more repetitions of the same K steps for median / min, then - `sustain_s` > 0, the headline workload - ONE
    contiguous region of about that many seconds of the same steps (K steps of a sub-millisecond product last 12 ms, too
    short for an outside utilisation sampler to ever see the device busy)."""
    mx = wl.mx
    for i in range(steps):  # create every hipEvent before the timed region (creation on the host would be timed)
        for j in range(wl.nmarks):
            wl.mark(i, j)
    for i in range(warmup):
        wl.step(i, False)
    if warmup == 0 and wl.gather is not None:
        # --warmup 0 on a sharded run: the exchange path (staging buffers, RCCL on the engine's stream) is still exercised
        # once outside the clock - its first use is set-up, not throughput
        wl.step(0, False)
    wl.drain()
    wl.ctx.marker(1)  # delimits the timed region in a profiler's dispatch list (tools/pmc_window.py); outside the clock
    d.barrier_sync(mx.gpu_device_sync)
    t0 = time.perf_counter()
    for i in range(steps):
        wl.step(i, True)
    wl.drain()
    d.barrier_sync(mx.gpu_device_sync)
    elapsed = d.max_over_ranks(time.perf_counter() - t0)
    wl.ctx.marker(2)
    kernel_ms = wl.kernel_ms(steps)
    reps = []
    for _ in range(repeats):
        d.barrier_sync(mx.gpu_device_sync)
        t1 = time.perf_counter()
        for i in range(steps):
            wl.step(i, False)
        wl.drain()
        d.barrier_sync(mx.gpu_device_sync)
        reps.append(d.max_over_ranks(time.perf_counter() - t1) * 1e3 / steps)
    sustained = None
    if sustain_s > 0.0:
        count = max(steps, int(sustain_s / max(statistics.median([elapsed / steps] + [r * 1e-3 for r in reps]), 1e-6)))
        count = int(d.max_over_ranks(float(count)))  # the same number of steps (and gathers) on every rank
        d.barrier_sync(mx.gpu_device_sync)
        t1 = time.perf_counter()
        for i in range(count):
            wl.step(i, False)
        wl.drain()
        d.barrier_sync(mx.gpu_device_sync)
        dt = d.max_over_ranks(time.perf_counter() - t1)
        sustained = {"steps": count, "seconds": dt, "ms_per_step": dt * 1e3 / count, "value": wl.units * count / dt}
    other = None
    if wl.gather is not None and hasattr(wl, "lazy_gather"):
        # the same K steps under the other exchange policy (per step <-> once per region), so that the line shows what the
        # choice is worth; `value` above is the policy named in the top-level "gather" field
        wl.lazy_gather = not wl.lazy_gather
        wl.step(0, False)
        wl.drain()
        d.barrier_sync(mx.gpu_device_sync)
        t2 = time.perf_counter()
        for i in range(steps):
            wl.step(i, False)
        wl.drain()
        d.barrier_sync(mx.gpu_device_sync)
        dt = d.max_over_ranks(time.perf_counter() - t2)
        other = {"gather": "lazy" if wl.lazy_gather else "step", "ms_per_step": dt * 1e3 / steps, "value": wl.units * steps / dt}
        wl.lazy_gather = not wl.lazy_gather
        wl.step(0, False)
        wl.drain()
    # every rank validates (its own block and every gathered foreign block); the verdict is collective so that one rank's
    # failure fails the whole job instead of leaving the others in the next barrier
    failure = None
    try:
        wl.check()
    except Exception as e:  # noqa: BLE001 - any failure (a GpuPolyError, an allocation) must still reach the collective below
        failure = e
    all_ok = d.min_over_ranks(0.0 if failure else 1.0)
    if failure is not None:
        raise failure
    if all_ok < 1.0:
        raise AssertionError(f"rank {d.rank}: another rank's self-validation failed")
    all_ms = [elapsed * 1e3 / steps] + reps
    return {
        "other_gather": other,
        "sustained": sustained,
        "elapsed_s": elapsed,
        "ms_per_step": elapsed * 1e3 / steps,
        "value": wl.units * steps / elapsed,
        "kernel_ms": kernel_ms,
        "repeats": {"count": len(all_ms), "steps_each": steps, "median_ms_per_step": statistics.median(all_ms),
                    "min_ms_per_step": min(all_ms), "max_ms_per_step": max(all_ms),
                    "value_at_median": wl.units / (statistics.median(all_ms) * 1e-3)},
    }


def roofline_of(wl: Workload, kernel_ms):
    """The marked kernel with the largest hipEvent time that has an algorithmic byte count."""
    cands = [(ms, label, algo) for (label, algo), ms in zip(wl.kernels, kernel_ms) if algo]
    if not cands:
        label, ms = wl.kernels[0][0], kernel_ms[0]
        return {"bound": "hbm", "kernel": label, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "kernel_ms": round(ms, 4)}
    ms, label, algo = max(cands)
    label = getattr(wl, "launched_kernel", None) or label
    achieved = algo / (ms * 1e-3) / 1e9
    traffic = source = None
    # PMC counters need their own rocprofv3 passes (the guide's HBM section): the figure is read from the committed record
    # of the same command (tools/collect_r05.sh -> tools/pmc_window.py: only the launches between bench.py's region markers
    # are counted, FETCH_SIZE doubled as the guide prescribes for gfx950), not measured in this run - and only while the
    # record's ISA hash of the kernel equals the hash of the kernel in the library loaded now (load_counters)
    rec = load_counters(wl.name) if wl.d.world == 1 else None
    krec = (rec or {}).get("kernels", {}).get(kernel_base(label))
    stale = None
    if krec is not None:
        stale = bool(krec["stale"])
        if not stale and krec.get("hbm_bytes_per_launch"):
            traffic = krec["hbm_bytes_per_launch"]
            source = f"{rec['file']} (head {rec.get('head')}, kernel {kernel_base(label)} isa {krec.get('isa_hash')}: " \
                     f"{rec.get('source', 'rocprofv3 --pmc passes')}; {krec.get('launches_per_step')} launch(es) per step)"
        elif stale:
            source = f"{rec['file']} was counted on another build of {kernel_base(label)} (isa {krec.get('isa_hash')}): not used"
    return {"bound": "hbm", "kernel": label, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source, "counters_stale": stale,
            "algorithmic_bytes_per_launch": algo, "kernel_ms": round(ms, 5)}


# ---------------------------------------------------------------------------------------------------
# composed roofline of a multi-kernel call (a preimage, a decomposition, a chain step)
# ---------------------------------------------------------------------------------------------------
SIMDS = 256 * 4          # MI355X: 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4          # peak engine clock (MI355X_MICROARCH.md); the chip holds less under load, so issue floors are optimistic
DEFAULT_VALU_CYCLES = 4.0  # a wave64 VALU instruction on a 16-lane SIMD; per-kernel prices from the ISA mix where known


def kernel_base(name: str) -> str:
    """`void ntt14::fwd_kernel<unsigned int, false>(unsigned int*, ...)` / `(ntt14::fwd_kernel<W, TIGHT>)` -> `ntt14::fwd_kernel`"""
    s = name.strip().strip("()").strip()
    if s.startswith("void "):
        s = s[5:]
    for ch in "<(":
        k = s.find(ch)
        if k > 0:
            s = s[:k]
    return s.strip()


def traced_kernels(wl: Workload, steps: int = 3):
    """Per-kernel hipEvent durations of `steps` steps of the workload: the library brackets every launch (and device copy)
    with two events on the stream it is enqueued on (gpupoly_trace_begin / _end).  Returns {kernel: {launches, ms, bytes}}
    per STEP, kernels in first-launch order."""
    from mxx_amd import _ffi

    wl.step(0, False)
    wl.drain()
    wl.mx.gpu_device_sync()
    _ffi.trace_begin()
    try:
        for i in range(steps):
            wl.step(i, False)
        wl.drain()
        wl.mx.gpu_device_sync()
    finally:
        entries = _ffi.trace_end()
    agg = {}
    for e in entries:
        a = agg.setdefault(kernel_base(e["kernel"]), {"launches": 0.0, "ms": 0.0, "bytes": 0.0})
        a["launches"] += 1.0 / steps
        a["ms"] += e["ms"] / steps
        a["bytes"] += e["bytes"] / steps
    return agg


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def load_counters(workload: str, current_hashes=None):
    """profiles/pmc_<workload>.json (tools/pmc_window.py) with a verdict per kernel: `stale` = the record was counted on a
    build whose ISA text of that kernel differs from the library's today (or carries no hash at all, as round 4's files) -
    its instruction counts and traffic are then NOT used.  None if there is no record."""
    for name in (f"pmc_{workload}.json", f"r04_pmc_{workload}.json"):
        rec = load_profile_json(name)
        if rec is not None:
            break
    else:
        return None
    if current_hashes is None:
        from mxx_amd import codeobj

        current_hashes = codeobj.kernel_isa_hashes()
    rec["file"] = f"profiles/{name}"
    versioned = any(k.get("isa_hash") for k in (rec.get("kernels") or {}).values())
    for base, k in (rec.get("kernels") or {}).items():
        if versioned and base.startswith("__amd_rocclr"):
            k["stale"] = False  # the runtime's own copy / fill kernels: not part of libgpupoly, nothing to compare
        else:
            k["stale"] = k.get("isa_hash") is None or k["isa_hash"] != current_hashes.get(base)
    rec["stale_kernels"] = sorted(b for b, k in (rec.get("kernels") or {}).items() if k["stale"])
    return rec


def load_valu_mix(current_hashes=None):
    """{kernel: cycles per VALU instruction} from profiles/valu_mix.json (tools/valu_mix.py), entries of other builds dropped"""
    rec = load_profile_json("valu_mix.json")
    if rec is None:
        return {}
    if current_hashes is None:
        from mxx_amd import codeobj

        current_hashes = codeobj.kernel_isa_hashes()
    return {b: k for b, k in (rec.get("kernels") or {}).items() if k.get("isa_hash") and k["isa_hash"] == current_hashes.get(b)}


def composed_roofline(wl: Workload, call_ms: float, steps: int = 3):
    """roofline of a call that is a SEQUENCE of kernels: per kernel the larger of its HBM byte floor (stated algorithmic
    bytes of its operands at the 8 TB/s peak) and its VALU issue floor (wave-level VALU instructions, counted by a committed
    rocprofv3 --pmc SQ_INSTS_VALU pass of this workload, at the ISA mix's cycles per instruction on 1024 SIMDs at 2.4 GHz);
    frac = sum of the floors / the untraced call time.  Durations per kernel are measured HERE with hipEvents."""
    agg = traced_kernels(wl, steps)
    pmc = load_counters(wl.name) or {}
    mix = load_valu_mix()
    pk = pmc.get("kernels", {})
    rows, floor_sum, useful_sum, byte_sum, byte_floor_sum, traced_ms, stale = [], 0.0, 0.0, 0.0, 0.0, 0.0, []
    for base, a in agg.items():
        byte_ms = a["bytes"] / (HBM_PEAK_GBS * 1e9) * 1e3
        rec = pk.get(base)
        valu = issue_ms = price = lanes = None
        if rec and rec["stale"]:
            stale.append(base)
        elif rec and rec.get("launches_per_step"):
            # the counted step and the traced step launch the same kernels; scale if the launch counts differ (other column count)
            valu = rec["SQ_INSTS_VALU"] * (a["launches"] / rec["launches_per_step"])
            price = (mix.get(base) or {}).get("cycles_per_inst", DEFAULT_VALU_CYCLES)
            issue_ms = valu * price / (SIMDS * CLOCK_GHZ * 1e9) * 1e3
            lanes = rec.get("lane_utilisation")
        floor = max(byte_ms, issue_ms or 0.0)
        # the issue floor prices the kernel's OWN executed instructions; lanes masked off inside them did no work, so the
        # useful part of an issue-bound floor is its lane-utilised share (never below the byte floor)
        useful = max(byte_ms, (issue_ms or 0.0) * (lanes if lanes is not None else 1.0))
        floor_sum += floor
        useful_sum += useful
        byte_sum += a["bytes"]
        byte_floor_sum += byte_ms
        traced_ms += a["ms"]
        rows.append({"kernel": base, "launches": round(a["launches"], 2), "ms": round(a["ms"], 4),
                     "algorithmic_bytes": a["bytes"] or None, "byte_floor_ms": round(byte_ms, 4),
                     "SQ_INSTS_VALU": round(valu) if valu else None, "cycles_per_inst": price,
                     "issue_floor_ms": round(issue_ms, 4) if issue_ms is not None else None,
                     "lane_utilisation": lanes,
                     "bound": "valu" if (issue_ms or 0.0) > byte_ms else "hbm",
                     "frac": round(floor / a["ms"], 4) if a["ms"] > 0 else None})
    rows.sort(key=lambda r_: -r_["ms"])
    achieved = byte_sum / (call_ms * 1e-3) / 1e9
    have_lanes = any(r_["lane_utilisation"] is not None for r_ in rows)
    weighted = sum((r_["issue_floor_ms"] or 0.0) for r_ in rows if r_["lane_utilisation"] is not None)
    fresh = bool(pk) and not stale
    return {"bound": "composed: per kernel max(HBM byte floor at 8 TB/s, VALU issue floor at the ISA mix's price on 1024 SIMDs x 2.4 GHz)",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(floor_sum / call_ms, 4), "frac_hbm_bytes_only": round(byte_floor_sum / call_ms, 4),
            # frac prices the instructions the kernels EXECUTE; frac_useful discounts the issue-bound floors by the counted
            # active-lane share of those instructions (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU) per kernel)
            "frac_useful": round(useful_sum / call_ms, 4) if have_lanes else None,
            "lane_utilisation": round(sum((r_["issue_floor_ms"] or 0.0) * r_["lane_utilisation"] for r_ in rows
                                          if r_["lane_utilisation"] is not None) / weighted, 4) if have_lanes and weighted > 0 else None,
            "counters_stale": (not fresh) if pk else None, "stale_kernels": stale,
            "counters_head": pmc.get("head"), "counters_file": pmc.get("file"),
            "traffic": pmc.get("hbm_bytes_per_step") if fresh else None,
            "traffic_source": pmc.get("source") if fresh else None,
            "algorithmic_bytes_per_call": byte_sum, "call_ms": round(call_ms, 4),
            "sum_of_kernel_ms_traced": round(traced_ms, 4), "sum_of_floors_ms": round(floor_sum, 4),
            "valu_counts_source": pmc.get("source") if pk else None,
            "kernels": rows[:14], "kernels_not_listed": max(0, len(rows) - 14)}


def block_json(wl: Workload, res, d: Dist, args, steps, warmup):
    return {
        "metric": wl.metric,
        "value": res["value"],
        "unit": wl.unit,
        "n_gpus": d.world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        # the mode of the N-sweep this line belongs to (the same at N = 1): "strong" where the workload splits one problem
        "scaling": args.scaling if isinstance(wl, (MatMul, Preimage)) else "weak",
        # sharded runs: "step" = the column blocks are all-gathered after EVERY step (overlapped with the next one), "lazy" =
        # once per timed region; null where nothing is exchanged (N = 1, weak scaling, independent batches)
        "gather": args.gather if getattr(wl, "strong", False) and isinstance(wl, (MatMul, Preimage)) else None,
        "other_gather": res.get("other_gather"),
        "launch": ("one process, N contexts (gpupoly_comm)" if d.inproc else "one process per GPU (torch.distributed)") if d.world > 1 or d.active else "single process",
        "vs_baseline": None,
        "dtype": "u32" if wl.word == 4 else "u64",
        "data": "synthetic",
        "config": {"workload": wl.desc, "ring_dim": wl.params.ring_dimension(), "limbs": wl.depth,
                   "limb_bits": 24 if wl.word == 4 else 51, "units_per_step": wl.units, "sharding": wl.sharding},
        "repeats": res["repeats"],
        **({"sustained": res["sustained"]} if res.get("sustained") else {}),
        "roofline": roofline_of(wl, res["kernel_ms"]),
        **({"kernel_launches_per_step": int(wl.launches_per_step)} if hasattr(wl, "launches_per_step") else {}),
        **({"exchange": torch_exchange_report(wl, d)} if d.active and not d.inproc and getattr(wl, "strong", False)
           and isinstance(wl, (MatMul, Preimage)) else {}),
    }


def kernels_block(m1: Workload, res):
    def entry(label, ms, algo):
        gbs = algo / (ms * 1e-3) / 1e9
        return {"kernel": label, "us": round(ms * 1e3, 2), "algorithmic_bytes": algo, "achieved_GBps": round(gbs, 1),
                "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}

    out = {"workload": m1.desc}
    (fl, fa), (gl, ga) = m1.kernels
    out["ntt_forward"] = entry(fl, res["kernel_ms"][0], fa)
    out["mul_intt_fused"] = entry(gl, res["kernel_ms"][1], ga)
    rec = load_counters("m1") if m1.d.world == 1 else None
    for key, base in (("ntt_forward", "ntt14::fwd_kernel"), ("mul_intt_fused", "ntt14::inv_kernel")):
        krec = (rec or {}).get("kernels", {}).get(base)
        if krec is not None:
            out[key]["counters_stale"] = bool(krec["stale"])
        if krec and not krec["stale"] and krec.get("hbm_bytes_per_launch"):  # counted HBM bytes per launch (committed rocprofv3 --pmc passes of `--workload m1`)
            out[key]["traffic"] = krec["hbm_bytes_per_launch"]
            out[key]["traffic_source"] = f"{rec['file']} (head {rec.get('head')})"
            if krec.get("SQ_WAIT_ANY") and krec.get("SQ_WAVE_CYCLES"):
                out[key]["SQ_WAIT_ANY_over_SQ_WAVE_CYCLES"] = round(krec["SQ_WAIT_ANY"] / krec["SQ_WAVE_CYCLES"], 3)
    mul_ms, inv_ms = m1.standalone_kernels()
    out["mod_mul"] = entry("elementwise_kernel<u32,mul,bcast> (pointwise mod-mul by a resident ring element, standalone)", mul_ms, m1.vec_bytes)
    out["ntt_inverse"] = entry("ntt14::inv_kernel<u32,signed> (inverse negacyclic NTT, standalone)", inv_ms, m1.vec_bytes)
    out["step_ms"] = res["ms_per_step"]
    out["ring_mults_per_s"] = res["value"]
    return out


def exchange_report(mx, wls, ranks_seen, backend, devices):
    """What the multi-GPU exchange of this line actually was: how many ranks the communicator holds, which backend moved
    the blocks, whether the devices can address each other, and that every rank validated every foreign block."""
    from mxx_amd import _ffi

    checked = [getattr(w, "foreign_blocks_checked", None) for w in wls]
    try:
        peer = _ffi.peer_access_matrix(sorted(set(devices)))
    except Exception as e:  # noqa: BLE001 - a report, not a gate
        peer = f"unavailable: {e}"
    ok = all(c is not None for c in checked)
    distinct = len(set(devices)) == len(devices)
    return {"ranks_seen": ranks_seen, "comm_backend": backend, "devices": devices, "peer_access": peer,
            "foreign_blocks_checked_per_rank": checked, "self_validated": ok, "distinct_devices": distinct,
            # no multi-device run of this exchange exists in the repo's history (the pool hands out one GPU per
            # box); a line with distinct devices and self_validated = true is the first evidence, and carries it itself
            "comm_verified_on_distinct_devices_before_this_run": False,
            "comm_verified_by_this_run": bool(ok and distinct and len(devices) > 1)}


def torch_exchange_report(wl, d):
    """one process per GPU: every rank reports its device and how many foreign blocks it validated (collective)"""
    from mxx_amd import _ffi

    mine = {"rank": d.rank, "device": wl.device, "foreign_blocks_checked": getattr(wl, "foreign_blocks_checked", None)}
    everyone = d.gather_objects(mine)
    devices = [e["device"] for e in everyone]
    try:
        peer = _ffi.peer_access_matrix(sorted(set(devices)))
    except Exception as e:  # noqa: BLE001
        peer = f"unavailable: {e}"
    checked = [e["foreign_blocks_checked"] for e in everyone]
    ok = all(c is not None for c in checked)
    distinct = len(set(devices)) == len(devices)
    return {"ranks_seen": d.dist.get_world_size(), "comm_backend": f"torch.distributed '{d.backend}'" + (" (RCCL)" if d.backend == "nccl" else " (rehearsal, not xGMI)"),
            "devices": devices, "peer_access": peer, "foreign_blocks_checked_per_rank": checked,
            "self_validated": ok, "distinct_devices": distinct,
            "comm_verified_on_distinct_devices_before_this_run": False,
            "comm_verified_by_this_run": bool(ok and distinct and len(devices) > 1 and d.backend == "nccl")}


def run_block_inproc(mx, wls, comm, steps, warmup, repeats):
    """run_block for N contexts of one process: a worker thread per context issues that context's steps (the ABI calls
    release the GIL and never block), a barrier per step lets one thread enqueue the step's all-gather through the C ABI
    once every context has enqueued its block; ONE host clock around the region, every device synchronised on both sides."""
    import threading

    n = len(wls)
    fulls = {}
    # what each worker hands to the exchange of step i: published BEFORE the step's barrier, in the slot i & 1, so the
    # gathering thread never reads a workload's live attributes while its worker is already inside step i + 1 (ADVICE r3:
    # MatMul re-points self.out, Preimage replaces self.x).  Slot i & 1 is written again only in step i + 2, which no
    # worker can reach before rank 0 has passed the barrier of step i + 1, i.e. has finished enqueueing the gather of step i.
    published = [[None] * n, [None] * n]

    def gather(slot, blocks):
        fulls[slot] = comm.all_gather_columns(list(blocks), fulls.get(slot))
        for wl, f in zip(wls, fulls[slot]):
            wl.full = f

    def region(k, mark):
        per_step = comm is not None and not wls[0].lazy_gather
        start, step_barrier, errors = threading.Barrier(n + 1), threading.Barrier(n), []

        def body(r):
            try:
                start.wait()
                for i in range(k):
                    wls[r].step(i, mark)
                    if per_step:
                        published[i & 1][r] = wls[r].local_block()
                        step_barrier.wait()
                        if r == 0:
                            gather(i & 1, published[i & 1])
            except BaseException as e:  # noqa: BLE001 - reported by the main thread
                errors.append(e)
                step_barrier.abort()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(n)]
        for t in threads:
            t.start()
        mx.gpu_device_sync()
        t0 = time.perf_counter()
        start.wait()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        if comm is not None and not per_step and k:
            gather(0, [wl.local_block() for wl in wls])  # the region's one exchange step: every worker has joined
        mx.gpu_device_sync()
        return time.perf_counter() - t0

    for wl in wls:
        for i in range(steps):
            for j in range(wl.nmarks):
                wl.mark(i, j)
    region(max(warmup, 1), False)
    elapsed = region(steps, True)
    kernel_ms = wls[0].kernel_ms(steps)
    reps = [region(steps, False) * 1e3 / steps for _ in range(repeats)]
    other = None
    if comm is not None:
        for wl in wls:
            wl.lazy_gather = not wl.lazy_gather
        region(1, False)
        dt = region(steps, False)
        other = {"gather": "lazy" if wls[0].lazy_gather else "step", "ms_per_step": dt * 1e3 / steps, "value": wls[0].units * steps / dt}
        for wl in wls:
            wl.lazy_gather = not wl.lazy_gather
        region(1, False)
    for wl in wls:
        wl.check()
    all_ms = [elapsed * 1e3 / steps] + reps
    return {"other_gather": other, "elapsed_s": elapsed, "ms_per_step": elapsed * 1e3 / steps, "value": wls[0].units * steps / elapsed,
            "kernel_ms": kernel_ms,
            "repeats": {"count": len(all_ms), "steps_each": steps, "median_ms_per_step": statistics.median(all_ms),
                        "min_ms_per_step": min(all_ms), "max_ms_per_step": max(all_ms),
                        "value_at_median": wls[0].units / (statistics.median(all_ms) * 1e-3)}}


def main_inproc(args, emit):
    import mxx_amd as mx
    from mxx_amd.parallel import GpuComm

    n, ndev = args.gpus, mx.detected_gpu_device_count()
    if ndev == 0:
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    share = ndev < n
    if share and os.environ.get("MXX_BENCH_INPROC_SHARE_DEVICES") != "1":
        raise SystemExit(f"bench.py --inproc: {n} contexts need {n} devices, {ndev} visible "
                         "(MXX_BENCH_INPROC_SHARE_DEVICES=1 rehearses the path with contexts sharing devices - not a measurement)")
    ranks = [InprocRank(n, r, 7000 + r if share else None) for r in range(n)]

    def run(name, repeats):
        wls = [WORKLOADS[name](mx, ranks[r], args, r % ndev) for r in range(n)]
        for wl in wls:
            wl.setup()
        comm = GpuComm([wl.params for wl in wls]) if wls[0].strong and isinstance(wls[0], (MatMul, Preimage)) else None
        res = run_block_inproc(mx, wls, comm, args.steps, args.warmup, repeats)
        line = block_json(wls[0], res, ranks[0], args, args.steps, args.warmup)
        line["config"]["comm_backend"] = comm.backend if comm is not None else None
        line["config"]["devices"] = [r % ndev for r in range(n)]
        line["exchange"] = exchange_report(mx, wls, len(comm) if comm is not None else n, comm.backend if comm is not None else None,
                                           [r % ndev for r in range(n)])
        if comm is not None:
            comm.close()
        return line

    if args.workload == "default":
        line = run("m2a", args.repeats)
        pre = run("m3a", min(args.repeats, 3))
        for key in ("n_gpus", "higher_is_better", "vs_baseline", "data", "scaling", "roofline"):
            pre.pop(key, None)
        line["preimage"] = pre
    else:
        line = run(args.workload, args.repeats)
    line["cpu_baseline"] = None  # an N = 1 leg (run `python bench.py`)
    emit(line)


# ---------------------------------------------------------------------------------------------------
# SURVEY 8 rows f1 (compact wire format) and f4 (gate batching) as driver-visible figures
# ---------------------------------------------------------------------------------------------------
def compact_bytes_block(mx, x, reps=5):
    """ABI-only store / load of a preimage (`gpu_matrix_store_compact_bytes` / `gpu_matrix_load_compact_bytes`,
    src/matrix/gpu_dcrt_poly.rs:956-1044 are their callers) into / from PINNED host memory: payload bytes, the device
    kernels' time from the library's launch trace against their HBM floor, and what is left of the synchronous call -
    the device-to-host copy - against PCIe.  `x` = an EVAL matrix (the M3A preimage); the store takes it to the
    coefficient domain in place, as the reference's does (MatrixSerde.cu:1108-1118), so every repetition starts from a clone."""
    import ctypes as C

    from mxx_amd import _ffi

    lib = _ffi.lib()
    p = x.params
    n, L = p.ring_dimension(), x.level + 1
    coeffs = x.nrow * x.ncol * n
    cap = (coeffs * sum(q.bit_length() for q in p.moduli()[:L]) + 7) // 8
    word = p.ctx().word_bytes()
    host = lib.gpu_pinned_alloc(cap)
    if not host:
        raise RuntimeError("compact_bytes_block: gpu_pinned_alloc failed")
    try:
        buf = C.cast(host, C.POINTER(C.c_uint8))
        bits, bpc, plen = C.c_uint16(0), C.c_uint16(0), C.c_size_t(0)
        store_ms, load_ms, kern_store, kern_load, rows = [], [], [], [], None
        for rep in range(reps + 1):
            m = x.clone()
            mx.gpu_device_sync()
            _ffi.trace_begin()
            t0 = time.perf_counter()
            _ffi.check_status(lib.gpu_matrix_store_compact_bytes(m.raw, buf, cap, C.byref(bits), C.byref(bpc), C.byref(plen)), "gpu_matrix_store_compact_bytes")
            dt = time.perf_counter() - t0
            ent = _ffi.trace_end()
            back = mx.GpuDCRTPolyMatrix(p, x.nrow, x.ncol, x.level, False)
            mx.gpu_device_sync()
            _ffi.trace_begin()
            t1 = time.perf_counter()
            _ffi.check_status(lib.gpu_matrix_load_compact_bytes(back.raw, buf, plen.value, bits.value), "gpu_matrix_load_compact_bytes")
            mx.gpu_device_sync()
            dl = time.perf_counter() - t1
            lent = _ffi.trace_end()
            if rep:  # the first repetition warms the kernels and the allocator
                store_ms.append(dt * 1e3)
                load_ms.append(dl * 1e3)
                kern_store.append(sum(e["ms"] for e in ent))
                kern_load.append(sum(e["ms"] for e in lent))
                rows = [{"kernel": kernel_base(e["kernel"]), "ms": round(e["ms"], 4)} for e in ent]
        m.is_ntt = False
        back.is_ntt = False
        assert back == m, "compact bytes: load(store(x)) != x"
    finally:
        lib.gpu_pinned_free(host)
    payload = plen.value
    s_ms, l_ms, ks, kl = (statistics.median(v) for v in (store_ms, load_ms, kern_store, kern_load))
    mat_bytes = float(x.nrow * x.ncol * L * n * word)
    # store: inverse transform (read + write), width pass (read), pack pass (read) + payload zeroed and written
    algo_store = 4.0 * mat_bytes + 2.0 * payload
    floor_ms = algo_store / (HBM_PEAK_GBS * 1e9) * 1e3
    d2h = max(s_ms - ks, 1e-6)
    return {"metric": "compact_store_payload_bytes_per_s", "value": payload / (s_ms * 1e-3), "unit": "B/s", "ms_per_step": s_ms,
            "config": {"workload": f"compact wire format of the M3A preimage ({x.nrow}x{x.ncol}, n=2^14, L={L}, EVAL in): ABI-only store "
                                   "into pinned host memory, then load back; no host-mirror framing"},
            "payload_bytes": payload, "max_coeff_bits": bits.value, "kernel_ms": ks, "d2h_ms": d2h,
            "pcie_GBps": payload / (d2h * 1e-3) / 1e9, "host_ms": 0.0,
            "load_ms": l_ms, "load_kernel_ms": kl, "store_kernels": rows,
            "roofline": {"bound": "hbm", "frac": round(floor_ms / ks, 4) if ks > 0 else None,
                         "algorithmic_bytes_per_call": algo_store, "byte_floor_ms": round(floor_ms, 4), "kernel_ms": round(ks, 4),
                         "note": "frac = HBM floor of the store's device kernels (INTT r+w, width pass r, pack pass r, payload "
                                 "memset + w) / their traced time; the call itself is bound by the D2H copy (pcie_GBps)"}}


def gate_batch_block(mx, device, count=16, reps=20):
    """one level of independent circuit gates as ONE call (`gpupoly_matrix_mul_batch`) against the loop
    of `gpu_matrix_mul` calls src/circuit/poly_circuit/eval.rs:269 issues, on the M4 ring (n=256, 12 x 51-bit limbs):
    `count` products (1x76)*(76x4)."""
    p = mx.GpuDCRTPolyParams(256, mx.gen_crt_basis(256, 12, 51), 17, gpu_ids=[device])
    ctx = p.ctx()
    us = mx.GpuDCRTPolyUniformSampler()
    ls = [us.sample_uniform(p, 1, 76, mx.DistType.FinRingDist()) for _ in range(count)]
    rs = [us.sample_uniform(p, 76, 4, mx.DistType.FinRingDist()) for _ in range(count)]

    def timed(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(reps):
            mx.gpu_device_sync()
            t0 = time.perf_counter()
            out = fn()
            mx.gpu_device_sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts), out

    loop_ms, loop_out = timed(lambda: [l_ * r_ for l_, r_ in zip(ls, rs)])
    batch_ms, batch_out = timed(lambda: mx.GpuDCRTPolyMatrix.mul_batch(ls, rs))
    assert all(a == b for a, b in zip(loop_out, batch_out)), "gate batch: batched products differ from the loop's"
    algo = count * float(1 * 76 + 76 * 4 + 1 * 4) * 256 * 12 * ctx.word_bytes()
    return {"metric": "gate_products_per_s", "value": count / (batch_ms * 1e-3), "unit": "products/s", "ms_per_step": batch_ms,
            "config": {"workload": f"{count} independent products (1x76)*(76x4), n=256, L=12 (51-bit): gpupoly_matrix_mul_batch (one "
                                   "call) against a loop of gpu_matrix_mul"},
            "loop_ms": loop_ms, "speedup_vs_loop": loop_ms / batch_ms,
            "roofline": {"bound": "hbm", "frac": round(algo / (batch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_call": algo,
                         "note": "launch- and latency-bound: 12 MB of operands"}}


def mixed_keys_block(mx, device, keys=8, per_key=2, reps=8):
    """`preimage_batched_sharded` as the GGH15 evaluation calls it (src/lookup/ggh15/pubkey_gpu.rs:615-971): one call with
    requests against SEVERAL trapdoors of one context - `keys` trapdoors x `per_key` 4-column targets on the M4 ring.  Requests
    that share a key are sampled together (`preimage_many`); the key groups run on worker contexts of the same device
    (MXX_PREIMAGE_WORKERS, default 4), against one after another with MXX_PREIMAGE_WORKERS=1."""
    p = mx.GpuDCRTPolyParams(256, mx.gen_crt_basis(256, 12, 51), 17, gpu_ids=[device])
    sampler = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    us = mx.GpuDCRTPolyUniformSampler()
    pairs = [sampler.trapdoor(p, 2) for _ in range(keys)]
    reqs = []
    for j in range(keys * per_key):
        td, a = pairs[j % keys]
        reqs.append((j, p, td, a, us.sample_uniform(p, 2, 4, mx.DistType.FinRingDist())))

    def timed(workers):
        before = os.environ.get("MXX_PREIMAGE_WORKERS")
        os.environ["MXX_PREIMAGE_WORKERS"] = str(workers)
        try:
            out = None
            for _ in range(3):
                out = sampler.preimage_batched_sharded(reqs)
            ts = []
            for _ in range(reps):
                mx.gpu_device_sync()
                t0 = time.perf_counter()
                out = sampler.preimage_batched_sharded(reqs)
                mx.gpu_device_sync()
                ts.append((time.perf_counter() - t0) * 1e3)
            return statistics.median(ts), out
        finally:
            if before is None:
                os.environ.pop("MXX_PREIMAGE_WORKERS", None)
            else:
                os.environ["MXX_PREIMAGE_WORKERS"] = before

    one_ms, _ = timed(1)
    many_ms, out = timed(4)
    for (_, x), (_, _, _, a, t) in zip(out, reqs):
        assert a * x == t, "A*x != u"
    cols = 4 * len(reqs)
    return {"metric": "trapdoor_preimages_per_s", "value": cols / (many_ms * 1e-3), "unit": "preimages/s", "ms_per_step": many_ms,
            "config": {"workload": f"one preimage_batched_sharded call: {keys} trapdoors x {per_key} requests of 4 columns, n=256, L=12 (51-bit), "
                                   "d=2; key groups dealt to 4 worker contexts (streams) of the device by one host thread"},
            "requests_in_flight": len(reqs), "loop_ms": one_ms, "speedup_vs_loop": one_ms / many_ms,
            "roofline": {"bound": "hbm", "frac": None, "note": "launch- and latency-bound; the figure of merit is the speed-up over one stream"}}


# ---------------------------------------------------------------------------------------------------
# the short line (what the driver parses) built from the full record
# ---------------------------------------------------------------------------------------------------
SHORT_LINE_LIMIT = 4000  # bytes; the driver keeps an 8 KB tail of stdout (round 4's 30 KB line could not be parsed)


def sig(x, digits=5):
    """float rounded to `digits` significant digits (ints, None and strings pass through)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or x != x or x in (float("inf"), float("-inf")):
        return x
    from math import floor, log10

    r = round(x, digits - 1 - int(floor(log10(abs(x)))))
    return int(r) if abs(r) >= 10 ** digits else r


def _cpu_short(cb):
    if not cb:
        return None
    out = {"value": sig(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind", "port"),
           "label": "CPU restatement (oracle/, not OpenFHE)", "sample": (cb.get("sample") or "")[:60]}
    one = cb.get("one_core") or {}
    if one.get("value") is not None:
        out["one_core_value"] = sig(one["value"])
    return out


def _config_record(blk):
    """ONE short record of a block of the full line: time, throughput, roofline fractions, CPU figure"""
    if not blk:
        return None
    rf = blk.get("roofline") or {}
    cb = blk.get("cpu_baseline") or {}
    rec = {"ms_per_step": sig(blk.get("ms_per_step")), "value": sig(blk.get("value")), "unit": blk.get("unit"),
           "frac": rf.get("frac")}
    for key, short in (("frac_hbm_bytes_only", "frac_hbm_bytes_only"), ("frac_useful", "frac_useful"), ("lane_utilisation", "lanes")):
        if rf.get(key) is not None:
            rec[short] = rf[key]
    if rf.get("counters_stale"):  # only when true: the counted floors / traffic were NOT used
        rec["counters_stale"] = True
    if rf.get("traffic") and (rf.get("algorithmic_bytes_per_call") or rf.get("algorithmic_bytes_per_launch")):
        rec["traffic_x"] = sig(rf["traffic"] / (rf.get("algorithmic_bytes_per_call") or rf["algorithmic_bytes_per_launch"]), 4)  # counted HBM bytes / algorithmic bytes
    if blk.get("kernel_launches_per_step") is not None:
        rec["launches"] = blk["kernel_launches_per_step"]
    for key, short in (("requests_in_flight", "requests"), ("speedup_vs_one_request", "speedup"), ("payload_bytes", "payload_bytes"),
                       ("kernel_ms", "kernel_ms"), ("d2h_ms", "d2h_ms"), ("pcie_GBps", "pcie_GBps"), ("speedup_vs_loop", "speedup_vs_loop"),
                       ("loop_ms", "loop_ms"), ("vs_extension_sequence", "vs_extension_sequence")):
        if blk.get(key) is not None:
            rec[short] = sig(blk[key], 4)
    if cb.get("value") is not None:
        rec["cpu_value"], rec["cpu_cores"] = sig(cb["value"]), cb.get("cores")
    return rec


def short_line(full):
    """The driver-facing line: the contract's keys for the headline workload (M2A), its roofline and CPU baseline, and a flat
    `configs` map with one short record per BASELINE configuration.  Everything else stays in bench_detail.json."""
    if "metric" not in full:
        return full  # --dry-run and the like
    rf = full.get("roofline") or {}
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = sig(line["value"], 7), sig(line["ms_per_step"], 6)
    line["config"] = {"workload": (cfg.get("workload") or "")[:140], "units_per_step": cfg.get("units_per_step"),
                      "sharding": (cfg.get("sharding") or "")[:100]}
    if full.get("gather") is not None:
        line["gather"] = full["gather"]
    og = full.get("other_gather")
    if og:  # N > 1: the same K steps under the other exchange policy (per step <-> once per region)
        line["other_gather"] = {"gather": og.get("gather"), "value": sig(og.get("value")), "ms_per_step": sig(og.get("ms_per_step"))}
    rep = full.get("repeats") or {}
    if rep:
        line["median_ms_per_step"] = sig(rep.get("median_ms_per_step"), 6)
    if full.get("sustained"):
        line["sustained"] = {k: sig(v) for k, v in full["sustained"].items() if k in ("steps", "seconds", "ms_per_step")}
    line["roofline"] = {"bound": rf.get("bound") if rf.get("bound") in ("hbm", "mfma") else "hbm",
                        "kernel": (rf.get("kernel") or "")[:72], "achieved": rf.get("achieved"), "peak": rf.get("peak"),
                        "unit": rf.get("unit"), "frac": rf.get("frac"), "traffic": rf.get("traffic"),
                        "algorithmic_bytes_per_launch": rf.get("algorithmic_bytes_per_launch"), "kernel_ms": rf.get("kernel_ms")}
    if rf.get("counters_stale") is not None:
        line["roofline"]["counters_stale"] = rf["counters_stale"]
    line["cpu_baseline"] = _cpu_short(full.get("cpu_baseline"))
    configs = {}
    k = full.get("kernels")
    if k:
        rec = {"ms_per_step": sig(k.get("step_ms")), "value": sig(k.get("ring_mults_per_s")), "unit": "ring-ops/s"}
        for key, short in (("ntt_forward", "ntt"), ("ntt_inverse", "intt"), ("mod_mul", "mul"), ("mul_intt_fused", "mul_intt")):
            e = k.get(key)
            if e:
                rec[short + "_us"], rec[short + "_frac"] = e.get("us"), e.get("frac_of_hbm_peak")
        lb = (k.get("large_batch") or {}).get("ntt_forward")
        if lb:
            rec["ntt_frac_beyond_cache"] = lb.get("frac_of_hbm_peak")
        cb = k.get("cpu_baseline") or {}
        if cb.get("value") is not None:
            rec["cpu_value"], rec["cpu_cores"] = sig(cb["value"]), cb.get("cores")
        configs["m1_ntt_mul"] = rec
    m2b = full.get("m2b")
    if m2b:
        configs["m2b_product"] = _config_record(m2b)
        if (m2b.get("roofline") or {}).get("frac_mac_floor") is not None:
            configs["m2b_product"]["frac_mac_floor"] = m2b["roofline"]["frac_mac_floor"]
        configs["m2b_decompose"] = _config_record(m2b.get("decompose"))
        configs["m2b_mul_decompose"] = _config_record(m2b.get("mul_decompose"))
    for key, name in (("preimage", "m3a_preimage"), ("preimage_reference_sequence", "m3a_reference_sequence"),
                      ("preimage_m3b", "m3b_preimage"), ("chain_m4", "m4_chain"), ("chain_m4_batched", "m4_chain_batched"),
                      ("preimage_mixed_keys", "m4_mixed_keys"), ("compact_bytes", "compact_bytes"), ("gate_batch", "gate_batch")):
        if full.get(key):
            configs[name] = _config_record(full[key])
    s8 = (full.get("preimage") or {}).get("shard_of_8")
    if s8 and configs.get("m3a_preimage"):
        configs["m3a_preimage"]["strong_eff8_predicted"] = s8.get("predicted_strong_scaling_efficiency_at_8")
    iu = full.get("independent_units")
    if iu:
        line["independent_units"] = {"value": sig(iu.get("value")), "ms_per_step": sig(iu.get("ms_per_step")), "scaling": "weak"}
    piu = (full.get("preimage") or {}).get("independent_units")
    if piu and configs.get("m3a_preimage"):
        configs["m3a_preimage"]["independent_units_value"] = sig(piu.get("value"))
    line["configs"] = {name: rec for name, rec in configs.items() if rec}
    ex = full.get("exchange") or (full.get("preimage") or {}).get("exchange")
    if ex:
        line["exchange"] = {k_: ex.get(k_) for k_ in ("ranks_seen", "comm_backend", "self_validated", "distinct_devices",
                                                      "comm_verified_by_this_run")}
        pex = (full.get("preimage") or {}).get("exchange")
        if pex and pex is not ex:
            line["exchange"]["preimage_self_validated"] = pex.get("self_validated")
    line["detail"] = "bench_detail.json"
    # never exceed the limit: drop the optional parts, least important first
    for victim in ("detail", "sustained", "other_gather", "independent_units"):
        if len(json.dumps(line)) <= SHORT_LINE_LIMIT:
            break
        line.pop(victim, None)
    for name in list(line.get("configs", {})):
        if len(json.dumps(line)) <= SHORT_LINE_LIMIT:
            break
        rec = line["configs"][name]
        line["configs"][name] = {k_: rec[k_] for k_ in ("ms_per_step", "value", "unit", "frac") if k_ in rec}
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.inproc:
        sys.exit(self_launch(args))  # nothing GPU-related has been imported or called in this process

    # stdout carries exactly ONE JSON line: native libraries write banners to file descriptor 1 (RCCL prints its
    # version block there when a communicator is created), so everything else is sent to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        """stdout gets ONE short JSON line (< 4 KB: the driver keeps an 8 KB tail of stdout and parses its last line); the
        full record - per-kernel tables, sources, repeats - goes to bench_detail.json next to this file (and under
        gpurun_out/ when that exists, so that it travels back from a GPU box) and to stderr."""
        detail = json.dumps(obj)
        for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
            try:
                if os.path.isdir(os.path.dirname(path)):
                    with open(path, "w") as f:
                        f.write(detail + "\n")
            except OSError as e:
                print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
        print("bench.py detail: " + detail, file=sys.stderr)
        sys.stderr.flush()
        os.write(json_fd, (json.dumps(short_line(obj)) + "\n").encode())

    if args.inproc and "WORLD_SIZE" not in os.environ:
        return main_inproc(args, emit)
    d = Dist(args)
    if args.dry_run:
        ok = 1.0
        if d.active:
            t = d.torch.ones(1, dtype=d.torch.float64, device=d.device())
            d.dist.all_reduce(t)
            ok = float(t.item())
        if d.rank == 0:
            emit({"dry_run": True, "world": d.world, "all_reduce": ok, "backend": d.backend if d.active else None})
        d.finish()
        return
    if d.active and d.world != args.gpus and d.rank == 0:
        print(f"bench.py: note: --gpus {args.gpus} but the job has {d.world} ranks; reporting the job's world size", file=sys.stderr)

    import mxx_amd as mx

    if mx.detected_gpu_device_count() == 0:
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    device = d.local_rank if d.active else 0
    if d.active and d.backend == "gloo":
        # rehearsal of the sharded path with more ranks than GPUs (gloo moves device tensors; RCCL refuses two ranks per
        # device): ranks share the devices that exist.  Timings of such a run say nothing about scaling.
        device = d.local_rank % mx.detected_gpu_device_count()
    steps, warmup = args.steps, args.warmup

    def run(name, steps_, warmup_, repeats_, sustain_s=0.0):
        wl = WORKLOADS[name](mx, d, args, device)
        wl.setup()
        res = run_block(wl, d, steps_, warmup_, repeats_, sustain_s)
        return wl, res

    def independent_units(name):
        """N > 1, strong run: the same workload once more as the reference itself uses several devices - every rank works on
        its own full-shape units, nothing is exchanged (gate-parallel circuit evaluation, `preimage_batched_sharded`):
        weak scaling, reported next to the strong figure"""
        import copy

        wargs = copy.copy(args)
        wargs.scaling = "weak"
        w = WORKLOADS[name](mx, d, wargs, device)
        w.setup()
        r = run_block(w, d, steps, warmup, 0)
        return {"scaling": "weak", "value": r["value"], "ms_per_step": r["ms_per_step"], "units_per_step": w.units,
                "sharding": w.sharding}

    def sub_block(name, steps_, repeats_, composed=True, cpu=True):
        """One more BASELINE configuration inside the default line: its own value / ms_per_step / repeats / roofline
        (composed from the launch trace for multi-kernel calls) and CPU baseline; keys shared with the top level dropped."""
        w, r = run(name, steps_, warmup, repeats_)
        blk = block_json(w, r, d, args, steps_, warmup)
        for key in ("n_gpus", "higher_is_better", "vs_baseline", "data"):
            blk.pop(key, None)
        blk["call_ms_hipevents"] = round(r["kernel_ms"][0], 4)
        if composed and d.world == 1 and not args.no_trace:
            blk["roofline"] = composed_roofline(w, r["kernel_ms"][0])
        if cpu and d.rank == 0 and d.world == 1 and not args.no_cpu_baseline:
            blk["cpu_baseline"] = cpu_baseline(name, args.cpu_seconds / 2)
        return w, r, blk

    if args.workload == "default":
        wl, res = run("m2a", steps, warmup, args.repeats, args.sustain)
        line = block_json(wl, res, d, args, steps, warmup)
        line["output_buffers"] = ("the product writes into pre-allocated output matrices; the reference's timed `&left * &right` "
                                  "also creates its output (src/matrix/gpu_dcrt_poly.rs:1792-1815) - microseconds from the stream-ordered cache")
        del wl
        if d.world > 1 and args.scaling == "strong":
            line["independent_units"] = independent_units("m2a")
        pre_wl, pre_res, pre = sub_block("m3a", steps, min(args.repeats, 3), cpu=False)
        pre.pop("scaling", None)
        if d.world > 1 and args.scaling == "strong":
            del pre_wl
            pre["independent_units"] = independent_units("m3a")
            pre_wl = None
        if d.world == 1:
            # one rank's share at N = 8 (7 of the 50 columns): what column sharding can reach before any exchange - a
            # call of 7 columns is not 7/50 of a call of 50 (fixed costs, partly filled sampler waves)
            t7 = uniform_matrix(mx, pre_wl.params, 1, 7, 9, total_cols=50, col_start=0)
            call = lambda: pre_wl.sampler.preimage(pre_wl.params, pre_wl.td, pre_wl.pub, t7)
            for _ in range(3):
                call()
            times = []
            for _ in range(5):
                pre_wl.ctx.timer_start()
                call()
                times.append(pre_wl.ctx.timer_stop())
            ms7 = statistics.median(times)
            pre["shard_of_8"] = {"columns": 7, "call_ms": round(ms7, 4), "linear_share_ms": round(pre_res["kernel_ms"][0] * 7 / 50, 4),
                                 "predicted_strong_scaling_efficiency_at_8": round(pre_res["kernel_ms"][0] / (8 * ms7), 3)}
        line["preimage"] = pre
        if d.world == 1:
            line["compact_bytes"] = compact_bytes_block(mx, pre_wl.x)
        del pre_wl
        if d.world == 1:
            w_ref, r_ref, ref = sub_block("m3a_refseq", max(5, steps // 2), 2, composed=False, cpu=False)
            del w_ref
            ref.pop("scaling", None)
            ref["vs_extension_sequence"] = ref["ms_per_step"] / pre["ms_per_step"]
            line["preimage_reference_sequence"] = ref
        m1, m1_res = run("m1", steps, warmup, 0)
        line["kernels"] = kernels_block(m1, m1_res)
        del m1
        if d.world == 1:
            line["kernels"]["large_batch"] = large_batch_kernels(mx, device)
        # ---- the remaining BASELINE configurations: M2B product + decompose + mul_decompose, M3B, M4 ----
        m2b_wl, m2b_res, m2b = sub_block("m2b", steps, min(args.repeats, 3), composed=False, cpu=True)
        m2b["roofline"]["also"] = ("dense contraction, 5.3 MAC/B: bound by L2 -> LDS operand delivery, not HBM (DESIGN.md section 5); "
                                   "integer-MAC floor at 5.0 cycles per v_mad_u64_u32: "
                                   f"{64 ** 3 * N_RING * 8 * 5.0 / 64 / (SIMDS * CLOCK_GHZ * 1e9) * 1e3:.2f} ms")
        del m2b_wl
        if d.world == 1:
            _, _, dec = sub_block("m2b_decompose", max(2, steps // 4), 2)
            m2b["decompose"] = dec
            _, _, md = sub_block("m2b_mul_decompose", max(2, steps // 4), 2)
            m2b["mul_decompose"] = md
        line["m2b"] = m2b
        w3, r3, m3b = sub_block("m3b", steps, min(args.repeats, 3))
        del w3
        line["preimage_m3b"] = m3b
        w4, r4, m4 = sub_block("m4", steps, min(args.repeats, 3))
        del w4
        line["chain_m4"] = m4
        w4b, r4b, m4b = sub_block("m4_batched", steps, min(args.repeats, 3), cpu=False)
        del w4b
        m4b["requests_in_flight"] = M4Batched.requests
        m4b["speedup_vs_one_request"] = m4b["value"] / m4["value"]
        line["chain_m4_batched"] = m4b
        if d.world == 1:
            line["gate_batch"] = gate_batch_block(mx, device)
            line["preimage_mixed_keys"] = mixed_keys_block(mx, device)
        line["baseline_configs"] = {
            "configs[0] (plumbing, n=2^12, L=2, 4x4)": "CPU-runnable parity case: tests/test_gpu_surface.py + tests/test_oracle.py, not a bench line",
            "configs[1] (NTT / INTT / mod-mul, n=2^14, L=4, 1024 polys)": "kernels",
            "configs[2] (64x64 product + gadget decompose, n=2^14, L=8)": "m2b (product), m2b.decompose, m2b.mul_decompose",
            "configs[3] (bench_preimage shape, L=8)": "preimage_m3b (the reference bench's own L=10 shape: preimage)",
            "configs[4] (GGH15 mod-p chain parameters, n=256, 51-bit limbs)": "chain_m4 (one request at a time), chain_m4_batched (16 requests per call)",
            "SURVEY 8 f1 / f4 and the reference's own preimage call sequence": "compact_bytes, gate_batch, preimage_reference_sequence",
            "benches/bench_matrix_mul_gpu.rs shape (the metric's own)": "top level (M2A)"}
        if d.rank == 0 and d.world == 1 and not args.no_cpu_baseline:  # an N = 1 leg: the other ranks would wait for it
            line["cpu_baseline"] = cpu_baseline("m2a", args.cpu_seconds)
            line["preimage"]["cpu_baseline"] = cpu_baseline("m3a", args.cpu_seconds)
            line["kernels"]["cpu_baseline"] = cpu_baseline("m1", args.cpu_seconds / 2)
    else:
        wl, res = run(args.workload, steps, warmup, args.repeats, args.sustain if args.workload == "m2a" else 0.0)
        line = block_json(wl, res, d, args, steps, warmup)
        if args.workload == "m1":
            line["kernels"] = kernels_block(wl, res)
            if d.world == 1:
                del wl
                wl = None
                line["kernels"]["large_batch"] = large_batch_kernels(mx, device)
        elif d.world == 1 and not args.no_trace and not isinstance(wl, MatMul):
            line["roofline"] = composed_roofline(wl, res["kernel_ms"][0])
        if d.rank == 0 and d.world == 1 and not args.no_cpu_baseline:  # an N = 1 leg: the other ranks would wait for it
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds)
    if d.rank == 0:
        line.setdefault("cpu_baseline", None)
        emit(line)
    d.finish()


# ---------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N=1 semantics): the CPU restatement under oracle/, built -march=native on THIS box,
# timed inside C (CLOCK_MONOTONIC around the kernels, inputs generated in C), median of >= 5 repetitions after a
# warm-up, on all host cores of the box's share and on 1 core
# ---------------------------------------------------------------------------------------------------
def host_cores() -> int:
    cores = len(os.sched_getaffinity(0))
    try:  # the box's CPU share, not the host's thread count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(wl: str, budget_s: float):
    import ctypes as C

    import numpy as np

    from oracle import oracle as O

    O.use_native_build()  # -O3 -march=native -fopenmp for this host, in a temp dir
    lib = O.lib()
    cores = host_cores()
    depth = {"m1": 4, "m2a": 15, "m2b": 8, "m2b_decompose": 8, "m2b_mul_decompose": 8, "m3a": 10, "m3b": 8, "m3a_refseq": 10, "m4": 12, "m4_batched": 12}[wl]
    n = 256 if wl.startswith("m4") else N_RING
    moduli = O.gen_crt_basis(n, depth, 51 if wl.startswith("m4") else 24)
    mod = np.asarray(moduli, dtype=np.uint64)
    mp = mod.ctypes.data_as(C.POINTER(C.c_uint64))
    dp = C.POINTER(C.c_double)
    out = {"kind": "port", "cores": cores, "label": "CPU restatement (oracle/, not OpenFHE), gcc -O3 -march=native -fopenmp"}
    sig = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.c_int, C.c_int, dp, C.c_int, dp]
    if wl in ("m2a", "m2b"):
        r, k, c = (1, 30, 120) if wl == "m2a" else (8, 64, 8)
        lib.orc_bench_matmul.restype = C.c_int
        lib.orc_bench_matmul.argtypes = [C.c_size_t] * 3 + sig
        fn = lambda ra, sa, ro, so: lib.orc_bench_matmul(r, k, c, depth, n, mp, cores, ra, sa, ro, so)
        units, unit, width = r * k * c, "ring-ops/s", 1
        macs = units * n * depth
        reps_all, reps_one = 9, (3 if macs < 2e9 else 1)
        sample = f"the full ({r}x{k})*({k}x{c}) product at n=2^14, L={depth}" + ("" if wl == "m2a" else " (an 8x8 output block of the 64x64 product)")
    elif wl == "m1":
        polys = 128
        lib.orc_bench_ring_mul.restype = C.c_int
        lib.orc_bench_ring_mul.argtypes = [C.c_size_t] + sig
        fn = lambda ra, sa, ro, so: lib.orc_bench_ring_mul(polys, depth, n, mp, cores, ra, sa, ro, so)
        units, unit, width = polys, "ring-ops/s", 3
        reps_all, reps_one = 9, 5
        sample = f"{polys} of 1024 polys, same step (NTT, *w, INTT), Shoup butterflies + Barrett product"
    elif wl in ("m2b_decompose", "m2b_mul_decompose"):
        return cpu_baseline_decompose(O, wl, n, moduli, cores, out)
    else:
        return cpu_baseline_preimage(O, wl, n, moduli, cores, budget_s, out)
    sec_all = np.zeros(reps_all * width, dtype=np.float64)
    sec_one = np.zeros(reps_one * width, dtype=np.float64)
    rc = fn(reps_all, sec_all.ctypes.data_as(dp), reps_one, sec_one.ctypes.data_as(dp))
    assert rc == 0, "CPU baseline allocation failed"
    t_all = sec_all.reshape(reps_all, width).sum(axis=1)
    t_one = sec_one.reshape(reps_one, width).sum(axis=1)
    out.update({"value": units / float(np.median(t_all)), "unit": unit,
                "median_s": float(np.median(t_all)), "min_s": float(t_all.min()), "reps": reps_all,
                "one_core": {"value": units / float(np.median(t_one)), "cores": 1, "median_s": float(np.median(t_one)),
                             "min_s": float(t_one.min()), "reps": reps_one},
                "sample": sample + "; timed inside C (CLOCK_MONOTONIC), inputs generated in C, median after one warm-up"})
    if wl == "m1":
        per = sec_all.reshape(reps_all, width)
        out["phase_median_s"] = {"ntt": float(np.median(per[:, 0])), "mul": float(np.median(per[:, 1])), "intt": float(np.median(per[:, 2]))}
    return out


def cpu_baseline_decompose(O, wl, n, moduli, cores, out):
    """G^-1 of a 2 x 2 block of the 64 x 64 matrix (inverse transform, digit extraction, forward transform of the 64 digit
    polynomials) and, for mul_decompose, the product of an (8 x 32) block of S with them: the C / OpenMP kernels of the
    restatement, Python only sequences them."""
    import numpy as np

    def timed(fn, reps, threads):
        O.lib().orc_set_threads(threads)
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return np.asarray(ts)

    m = O.matrix_ntt(O.random_matrix(11, 2, 2, moduli, n), moduli)
    k = O.digits_per_tower(moduli, 12) * len(moduli)
    dec = lambda: O.matrix_ntt(O.decompose(O.matrix_ntt(m, moduli, inverse=True), moduli, 12), moduli)
    if wl == "m2b_decompose":
        fn, units, unit = dec, 4, "ring-elements/s"
        sample = "G^-1 of a 2 x 2 block of the 64 x 64 matrix (4 ring elements -> 64 digit polynomials in EVAL form)"
    else:
        s = O.matrix_ntt(O.random_matrix(12, 8, 2 * k, moduli, n), moduli)
        fn, units, unit = (lambda: O.matmul(s, dec(), moduli, fast=True)), 8 * 2 * k * 2, "ring-ops/s"
        sample = f"(8 x {2 * k}) * G^-1(2 x 2): a 2-column, 2-source-row block of the (8 x 1024) * G^-1(64 x 64) call"
    t_all, t_one = timed(fn, 7, cores), timed(fn, 3, 1)
    O.lib().orc_set_threads(cores)
    out.update({"value": units / float(np.median(t_all)), "unit": unit, "median_s": float(np.median(t_all)), "min_s": float(t_all.min()),
                "reps": 7, "one_core": {"value": units / float(np.median(t_one)), "cores": 1, "median_s": float(np.median(t_one)),
                                        "min_s": float(t_one.min()), "reps": 3},
                "sample": sample + "; median after a warm-up"})
    return out


def cpu_baseline_preimage(O, wl, n, moduli, cores, budget_s, out):
    """oracle.preimage: the whole chain (p2, p1, G-sampling, products, NTTs) with the trapdoor and the covariance
    factors prepared outside, as on the GPU; the kernels are C/OpenMP, Python only sequences them."""
    import numpy as np

    base, sigma = (17, 4.578) if wl.startswith("m4") else (12, 4.578)
    cols = 4
    seed = bytes(range(32))
    O.lib().orc_set_threads(cores)
    r, e, a = O.trapdoor_gen(moduli, n, base, sigma, 1, seed)
    _, c_par, s_par = O.preimage_params(moduli, n, base, sigma, 1)
    inv = lambda m: O.matrix_ntt(m, moduli, inverse=True)
    rt, et = np.swapaxes(r, 0, 1), np.swapaxes(e, 0, 1)
    cov = O.p1_covariance(inv(O.matmul(r, rt, moduli, fast=True)), inv(O.matmul(r, et, moduli, fast=True)),
                          inv(O.matmul(e, et, moduli, fast=True)), moduli, c_par, s_par, sigma)

    def timed(cols_, reps):
        target = O.matrix_ntt(O.random_matrix(7, 1, cols_, moduli, n), moduli)
        O.preimage(moduli, n, base, sigma, r, e, a, target, seed, cov=cov)  # warm-up
        ts = []
        for i in range(reps):
            t0 = time.perf_counter()
            O.preimage(moduli, n, base, sigma, r, e, a, target, bytes([i]) + seed[1:], cov=cov)
            ts.append(time.perf_counter() - t0)
        return np.asarray(ts)

    t_all = timed(cols, 5)
    O.lib().orc_set_threads(1)
    one_cols = 1
    t_one = timed(one_cols, 3)
    O.lib().orc_set_threads(cores)
    out.update({"value": cols / float(np.median(t_all)), "unit": "preimages/s", "median_s": float(np.median(t_all)),
                "min_s": float(t_all.min()), "reps": 5,
                "one_core": {"value": one_cols / float(np.median(t_one)), "cores": 1, "median_s": float(np.median(t_one)),
                             "min_s": float(t_one.min()), "reps": 3, "sample": f"{one_cols} target column per call"},
                "sample": f"{cols} of 50 target columns per call, same chain (p2, p1, G-sampling, products, NTTs), median of 5 calls after a warm-up"})
    return out


if __name__ == "__main__":
    main()
