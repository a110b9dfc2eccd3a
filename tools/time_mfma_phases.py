"""Phase timing of the matrix-core product (matmul_mfma.hip) at M2b: full kernel, without global loads,
loads + conversion only, without the byte split.  Run once per mode: MXX_HIP_MFMA_MODE=m python tools/time_mfma_phases.py
Needs a library built with `make -C mxx_amd/csrc PHASE_TIMING=1` (modes 1 and 2 produce wrong results by design and are
not compiled into the shipped library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L = 16384, 8
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
us = mx.GpuDCRTPolyUniformSampler()
a = us.sample_uniform(p, 64, 64, mx.DistType.FinRingDist())
b = us.sample_uniform(p, 64, 64, mx.DistType.FinRingDist())
out = mx.GpuDCRTPolyMatrix(p, 64, 64, L - 1, True)
ctx = p.ctx()
lib = _ffi.lib()
for path in (os.environ.get("PATHS", "mfma,dma")).split(","):
    os.environ["MXX_HIP_MATMUL_PATH"] = path
    _ffi.reload_env()
    for _ in range(3):
        lib.gpu_matrix_mul(out.raw, a.raw, b.raw)
    ctx.timer_start()
    for _ in range(10):
        lib.gpu_matrix_mul(out.raw, a.raw, b.raw)
    print(f"path={path} mode={os.environ.get('MXX_HIP_MFMA_MODE', '0')}: {ctx.timer_stop() / 10:.3f} ms per 64^3 product (L=8)")
