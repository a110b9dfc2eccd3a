"""Forward / inverse 2^14-point NTT time per vector against batch footprint (in-place transforms; the 268 MB M1 batch
half-lives in the 256 MiB Infinity Cache, the 34 GB digit matrix of a 64x64 decompose does not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L = 16384, 8
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
ctx, lib = p.ctx(), _ffi.lib()
for polys in (128, 512, 2048, 8192, 65536):
    m = mx.GpuDCRTPolyMatrix(p, polys, 1, L - 1, False)
    lib.gpu_matrix_ntt_all(m.raw); lib.gpu_matrix_intt_all(m.raw)
    mx.gpu_device_sync()
    f = i = 1e9
    for _ in range(3):
        ctx.timer_start(); lib.gpu_matrix_ntt_all(m.raw); f = min(f, ctx.timer_stop())
        ctx.timer_start(); lib.gpu_matrix_intt_all(m.raw); i = min(i, ctx.timer_stop())
    v = polys * L
    print(f"{v:7d} vectors ({v * n * 4 / 2**30:6.2f} GiB): forward {f * 1e6 / v:6.1f} ns/vector ({2 * v * n * 4 / f / 1e6:7.1f} GB/s), inverse {i * 1e6 / v:6.1f} ns/vector")
    del m
