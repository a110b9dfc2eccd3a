"""64-bit words, 51-bit limbs: the double-precision transforms (ntt_f64.h) against the integer ones (MXX_HIP_NTT64=int),
4 limbs, 4096 and 64 polys, n = 2^10..2^14; GB/s = 2 n w bytes per vector."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

lib = _ffi.lib()
us = mx.GpuDCRTPolyUniformSampler()
for logn, bits in ((10, 51), (11, 51), (12, 51), (13, 51), (14, 51), (15, 51), (16, 51), (17, 51), (16, 32), (14, 32)):
    n = 1 << logn
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 4, bits), 16)
    ctx = p.ctx()
    for polys in ((64, 4096) if logn <= 14 else (64, 512)):
        m = us.sample_uniform(p, polys, 1, mx.DistType.FinRingDist())
        line = f"{bits}-bit n=2^{logn} polys={polys:5d}:"
        for mode in ("int", "f64"):
            if mode == "int":
                os.environ["MXX_HIP_NTT64"] = "int"
            else:
                os.environ.pop("MXX_HIP_NTT64", None)
            _ffi.reload_env()
            res = []
            for inverse in (True, False):  # starts in EVAL
                fn = lib.gpu_matrix_intt_all if inverse else lib.gpu_matrix_ntt_all
                other = lib.gpu_matrix_ntt_all if inverse else lib.gpu_matrix_intt_all
                best = 1e9
                for _ in range(4):
                    ctx.timer_start(); _ffi.check_status(fn(m.raw), "ntt"); ms = ctx.timer_stop(); best = min(best, ms)
                    _ffi.check_status(other(m.raw), "ntt")
                _ffi.check_status(fn(m.raw), "ntt")  # leave the matrix in the format the next direction starts from
                res.append(best)
            nb = 2.0 * polys * 4 * n * 8
            line += f"  {mode}: inverse {res[0]:7.3f} ms {nb / res[0] / 1e6:7.1f} GB/s, forward {res[1]:7.3f} ms {nb / res[1] / 1e6:7.1f} GB/s;"
        print(line, flush=True)
        del m
