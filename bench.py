#!/usr/bin/env python3
"""bench.py — ring-ops/s of the DCRT hot path on MI355X, next to the roofline and the CPU path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload m1|m2a|m2b|m3a|m3b]

Default workload = BASELINE.json configs[1] ("M1"): n=2^14, 4 RNS limbs
(24-bit), batch of 1024 polynomials resident in HBM.  One step = one pass of the hot path
over the batch: x <- INTT( NTT(x) o w ) for all 1024 polynomials, i.e. 1024 ring
multiplications (ring-ops) by a resident EVAL-form ring element w, through the C ABI
(gpu_matrix_ntt_all, gpu_matrix_mul_scalar, gpu_matrix_intt_all).
Other workloads: m2a = the reference's benches/bench_matrix_mul_gpu.rs shape
(n=2^14, L=15, (1x30)*(30x120) = 3600 ring-ops/step); m2b = 64x64 * 64x64, L=8;
m3a = benches/bench_preimage_gpu.rs (n=2^14, L=10, sigma=4.578, d=1, 50 target columns;
unit preimages/s); m3b = BASELINE.json configs[3]: the same with L=8 - with N>1 every rank samples the preimages
of its own 50 target columns (the partition of preimage_batched_sharded) and the preimage blocks are all-gathered
over RCCL/xGMI, the one exchange step of that configuration.

N>1: one process per GPU (torch.distributed / RCCL only for the barrier and the
max-over-ranks clock); the path shards by independent polynomials / target columns with
no data-path collective, so every rank runs the same per-GPU batch ("scaling": "weak").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

BASIS_24 = None


def moduli_24(gen, n, depth):
    return gen(n, depth, 24)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="m1", choices=["m1", "m2a", "m2b", "m3a", "m3b"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


class Dist:
    """torch.distributed only when N>1 (importing torch costs minutes on a fresh box)."""

    def __init__(self, n_gpus: int):
        self.world = n_gpus
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = None
        if n_gpus > 1 or os.environ.get("MXX_BENCH_FORCE_DIST") == "1":  # the env rehearses the N>1 path on one GPU
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            self.torch, self.dist = torch, dist

    def barrier_sync(self, mx):
        mx.gpu_device_sync()
        if self.torch is not None:
            self.torch.cuda.synchronize()
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value: float) -> float:
        if self.torch is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        if self.torch is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def upload_random(mx, params, rows, cols, seed, eval_format, chunk_polys=256):
    """Fill a device matrix with i.i.d. uniform residues (splitmix64), chunked uploads."""
    from mxx_amd import _ffi

    n, moduli = params.ring_dimension(), params.moduli()
    L = len(moduli)
    out = mx.GpuDCRTPolyMatrix(params, rows, cols, L - 1, eval_format)
    q = np.asarray(moduli, dtype=np.uint64).reshape(1, L, 1)
    total = rows * cols
    rng = np.random.Generator(np.random.SFC64(seed))
    # upload row blocks through copy_block of a staging matrix to bound host memory
    flat = mx.GpuDCRTPolyMatrix(params, 1, min(chunk_polys, total), L - 1, eval_format)
    done = 0
    while done < total:
        cnt = min(chunk_polys, total - done)
        host = rng.integers(0, 1 << 63, size=(cnt, L, n), dtype=np.uint64) % q
        if cnt != flat.ncol:
            flat = mx.GpuDCRTPolyMatrix(params, 1, cnt, L - 1, eval_format)
        flat.load_rns(host.reshape(1, cnt, L, n), eval_format)
        # polys are row-major contiguous: copy as a 1 x cnt run when aligned to rows, else entry-wise
        p = done
        off = 0
        while off < cnt:
            r, c = divmod(p, cols)
            run = min(cols - c, cnt - off)
            out.copy_block_from(flat, r, c, 0, off, 1, run)
            out.is_ntt = eval_format
            p += run
            off += run
        done += cnt
    return out


def main():
    args = parse_args()
    # torch (N>1 only) must be imported BEFORE libgpupoly is loaded: the wheel bundles its own
    # libamdhip64.so.7, and a process must hold exactly one HIP runtime; loaded in this order
    # libgpupoly binds to the copy torch already mapped (same soname)
    d = Dist(args.gpus)
    import mxx_amd as mx
    from mxx_amd import _ffi

    if mx.detected_gpu_device_count() == 0:
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    device = d.local_rank if d.torch is not None else 0
    n = 16384
    wl = args.workload
    depth = {"m1": 4, "m2a": 15, "m2b": 8, "m3a": 10, "m3b": 8}[wl]
    moduli = mx.gen_crt_basis(n, depth, 24)
    params = mx.GpuDCRTPolyParams(n, moduli, 12, gpu_ids=[device])
    ctx = params.ctx()
    word = ctx.word_bytes()
    L = depth
    steps, warmup = args.steps, args.warmup
    roof = None
    extra = {}

    if wl == "m1":
        batch = 1024
        x = upload_random(mx, params, batch, 1, 0x6D7878 ^ 2, False)
        w = upload_random(mx, params, 1, 1, 0x6D7878 ^ 3, True)
        lib = _ffi.lib()

        def step(i, mark):
            if mark:
                ctx.timer_mark(2 * i)
            _ffi.check_status(lib.gpu_matrix_ntt_all(x.raw), "gpu_matrix_ntt_all")
            if mark:
                ctx.timer_mark(2 * i + 1)
            _ffi.check_status(lib.gpu_matrix_mul_scalar(x.raw, x.raw, w.raw), "gpu_matrix_mul_scalar")
            _ffi.check_status(lib.gpu_matrix_intt_all(x.raw), "gpu_matrix_intt_all")

        units_per_step = batch
        metric, unit = "dcrt_ring_ops_per_s", "ring-ops/s"
        kernel_name = "ntt14::fwd_kernel<uint32_t> (forward negacyclic NTT, 2^14 points)"
        algo_bytes = 2.0 * n * word * batch * L  # 2*n*w per (poly, limb)
        workload_desc = f"M1: n=2^14, L=4 (24-bit), batch {batch} polys; step = x<-INTT(NTT(x) o w) = {batch} ring mults"
    elif wl in ("m2a", "m2b"):
        r, k, c = (1, 30, 120) if wl == "m2a" else (64, 64, 64)
        a = upload_random(mx, params, r, k, 0x6D7878 ^ 4, True)
        b = upload_random(mx, params, k, c, 0x6D7878 ^ 5, True)
        out = mx.GpuDCRTPolyMatrix(params, r, c, L - 1, True)
        lib = _ffi.lib()

        gather = None
        if d.torch is not None:
            # the one real exchange step of the sharded product: every rank owns a block of
            # `c` output columns and all-gathers the blocks over xGMI (RCCL), zero-copy from
            # the engine's HBM allocation
            from mxx_amd.parallel import DeviceBuffer

            src_t = DeviceBuffer(out).tensor(device)
            full_t = d.torch.empty(d.dist.get_world_size() * src_t.numel(), dtype=src_t.dtype, device=src_t.device)
            gather = (src_t, full_t)

        def step(i, mark):
            if mark:
                ctx.timer_mark(2 * i)
            _ffi.check_status(lib.gpu_matrix_mul(out.raw, a.raw, b.raw), "gpu_matrix_mul")
            if mark:
                ctx.timer_mark(2 * i + 1)
            if gather is not None:
                mx.gpu_device_sync()  # engine stream -> torch stream hand-off
                d.dist.all_gather_into_tensor(gather[1], gather[0])

        units_per_step = r * k * c
        metric, unit = "dcrt_ring_ops_per_s", "ring-ops/s"
        kernel_name = "matmul_kernel<uint32_t,...> / mmdma::kernel_u32 (R_q matrix product, EVAL)"
        algo_bytes = float(r * k + k * c + r * c) * n * L * word
        workload_desc = f"{wl.upper()}: n=2^14, L={L} (24-bit), ({r}x{k})*({k}x{c}); 1 ring-op = one R_q multiply-accumulate"
    else:  # m3a / m3b
        sigma, dsize, cols = 4.578, 1, 50
        sampler = mx.GpuDCRTPolyTrapdoorSampler(params, sigma)
        td, pub = sampler.trapdoor(params, dsize)
        target = mx.GpuDCRTPolyUniformSampler().sample_uniform(params, dsize, cols, mx.DistType.FinRingDist())
        keep = {}
        gather_full = None
        if d.torch is not None and wl == "m3b":
            from mxx_amd.parallel import DeviceBuffer

            words = (params.modulus_digits() + 2) * dsize * cols * L * n  # one rank's preimage block
            gather_full = d.torch.empty(d.dist.get_world_size() * words * word, dtype=d.torch.uint8, device=device)

        def step(i, mark):
            if mark:
                ctx.timer_mark(2 * i)
            keep["x"] = sampler.preimage(params, td, pub, target)
            if mark:
                ctx.timer_mark(2 * i + 1)
            if gather_full is not None:
                mx.gpu_device_sync()  # engine stream -> torch stream hand-off
                d.dist.all_gather_into_tensor(gather_full, DeviceBuffer(keep["x"]).tensor(device))

        units_per_step = cols
        metric, unit = "trapdoor_preimages_per_s", "preimages/s"
        kernel_name = "preimage call (all kernels)"
        algo_bytes = None
        workload_desc = (f"{wl.upper()}: bench_preimage shape n=2^14, L={L}, base 2^12, sigma={sigma}, d=1, {cols} target columns"
                         + (" per rank, preimage blocks all-gathered (RCCL)" if gather_full is not None else ""))

    for i in range(warmup):
        step(i, False)
    d.barrier_sync(mx)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, True)
    d.barrier_sync(mx)
    elapsed = time.perf_counter() - t0
    elapsed = d.max_over_ranks(elapsed)

    if wl in ("m3a", "m3b"):
        x = keep["x"]
        assert pub * x == target, "A*x != u"

    result = None
    if d.rank == 0:
        ms_per_step = elapsed * 1e3 / steps
        value = units_per_step * steps * args.gpus / elapsed
        kernel_ms = float(np.mean([ctx.timer_elapsed(2 * i, 2 * i + 1) for i in range(steps)]))
        if algo_bytes is not None:
            achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{wl}.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms": round(kernel_ms, 5),
            }
        else:
            roof = {"bound": "hbm", "kernel": kernel_name, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None, "traffic": None, "kernel_ms": round(kernel_ms, 4)}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(wl, n, moduli, args.cpu_seconds)
        result = {
            "metric": metric,
            "value": value,
            "unit": unit,
            "n_gpus": args.gpus,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": workload_desc, "ring_dim": n, "limbs": L, "limb_bits": 24,
                       "units_per_step_per_gpu": units_per_step,
                       "sharding": ("column blocks per rank + RCCL all-gather of the result" if (wl in ("m2a", "m2b", "m3b") and args.gpus > 1)
                                    else "independent polys / target columns per rank, no collective")},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(result), flush=True)
    d.finish()


def cpu_baseline(wl, n, moduli, budget_s):
    """The CPU restatement (oracle/, kind 'port') on this box's host cores, bounded sample."""
    from oracle import oracle as O

    # the box's CPU share, not the host's thread count (a 1-GPU box gets a slice of the host)
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    O.lib().orc_set_threads(cores)
    L = len(moduli)
    if wl == "m1":
        polys = 128
        x = O.random_matrix(1, polys, 1, moduli, n)
        w = O.matrix_ntt(O.random_matrix(2, 1, 1, moduli, n), moduli)

        def run():
            y = O.matrix_ntt(x, moduli)
            y = O.pointwise("mul", y, w, moduli)
            return O.matrix_ntt(y, moduli, inverse=True)

        units, unit = polys, "ring-ops/s"
        sample = f"{polys} of 1024 polys, same step (NTT, *w, INTT), Shoup butterflies, OpenMP"
    elif wl in ("m2a", "m2b"):
        r, k, c = (1, 30, 120) if wl == "m2a" else (8, 64, 8)
        nn = n if wl == "m2a" else n
        a = O.random_matrix(1, r, k, moduli, nn)
        b = O.random_matrix(2, k, c, moduli, nn)

        def run():
            return O.matmul(a, b, moduli, fast=True)

        units, unit = r * k * c, "ring-ops/s"
        sample = f"({r}x{k})*({k}x{c}) at n=2^14, L={L}" + ("" if wl == "m2a" else " (8x8 output block of the 64x64 product)")
    else:  # m3a: the whole preimage chain (oracle.preimage), trapdoor and covariance factors prepared outside
        base, sigma, cols = 12, 4.578, 4
        seed = bytes(range(32))
        r, e, a = O.trapdoor_gen(moduli, n, base, sigma, 1, seed)
        _, c_par, s_par = O.preimage_params(moduli, n, base, sigma, 1)
        inv = lambda m: O.matrix_ntt(m, moduli, inverse=True)
        rt, et = np.swapaxes(r, 0, 1), np.swapaxes(e, 0, 1)
        cov = O.p1_covariance(inv(O.matmul(r, rt, moduli, fast=True)), inv(O.matmul(r, et, moduli, fast=True)),
                              inv(O.matmul(e, et, moduli, fast=True)), moduli, c_par, s_par, sigma)
        target = O.matrix_ntt(O.random_matrix(7, 1, cols, moduli, n), moduli)

        def run():
            return O.preimage(moduli, n, base, sigma, r, e, a, target, seed, cov=cov)

        units, unit = cols, "preimages/s"
        sample = f"{cols} of 50 target columns per call, same chain (p2, p1, G-sampling, products, NTTs), OpenMP"
    run()  # warm-up (tables, page faults)
    t0 = time.perf_counter()
    reps = 0
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 100000:
            break
    return {"value": units * reps / el, "unit": unit, "cores": cores, "kind": "port",
            "sample": sample + f"; {reps} reps in {el:.1f} s; CPU restatement (not OpenFHE)"}


if __name__ == "__main__":
    main()
