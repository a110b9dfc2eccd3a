"""Forward / inverse transform GB/s (2 n w bytes per vector) across ring sizes and batch sizes, u32 and u64 words."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

lib = _ffi.lib()
us = mx.GpuDCRTPolyUniformSampler()
for bits, L in ((24, 4), (51, 4)):
    for logn in (8, 10, 12, 13, 14, 15, 16, 17):
        n = 1 << logn
        p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, bits), 12)
        ctx = p.ctx()
        w = ctx.word_bytes()
        for polys in (1, 64, 4096):
            if polys * n * L * w > (2 << 30):
                continue
            m = us.sample_uniform(p, polys, 1, mx.DistType.FinRingDist())
            res = []
            for inverse in (True, False):  # starts in EVAL
                fn = lib.gpu_matrix_intt_all if inverse else lib.gpu_matrix_ntt_all
                other = lib.gpu_matrix_ntt_all if inverse else lib.gpu_matrix_intt_all
                best = 1e9
                for _ in range(3):
                    ctx.timer_start(); _ffi.check_status(fn(m.raw), "ntt"); ms = ctx.timer_stop(); best = min(best, ms)
                    _ffi.check_status(other(m.raw), "ntt")
                _ffi.check_status(fn(m.raw), "ntt")
                res.append(best)
            nb = 2.0 * polys * L * n * w
            print(f"{bits}-bit n=2^{logn:2d} polys={polys:5d}: inverse {res[0]:8.3f} ms {nb / res[0] / 1e6:8.1f} GB/s   forward {res[1]:8.3f} ms {nb / res[1] / 1e6:8.1f} GB/s", flush=True)
            del m
