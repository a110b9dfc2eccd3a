"""M1 with the host hand-over included: load_rns_batch (host u64 wire layout -> HBM), the step
(NTT, *w, INTT), store_rns_batch back.  Pageable numpy buffers and pinned buffers (gpu_pinned_alloc)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

n, L, polys = 16384, 4, 1024
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
rng = np.random.default_rng(1)
host = np.stack([rng.integers(0, q, size=(polys, 1, n), dtype=np.uint64) for q in p.moduli()], axis=2)  # (polys,1,L,n)
w = mx.GpuDCRTPolyMatrix.from_rns(p, host[:1], True)
lib = _ffi.lib()
nbytes = host.nbytes


def run(buf_in, buf_out, label):
    m = mx.GpuDCRTPolyMatrix(p, polys, 1, L - 1, False)
    best = 1e9
    for _ in range(4):
        mx.gpu_device_sync()
        t0 = time.perf_counter()
        ev = C.c_void_p()
        _ffi.check_status(lib.gpu_matrix_load_rns_batch(m.raw, buf_in, n * L * 8, 0, C.byref(ev)), "load")
        _ffi.wait_and_destroy_events(ev)
        m.is_ntt = False
        t1 = time.perf_counter()
        m.ntt_all_in_place()
        r = m.mul_scalar(w) if hasattr(m, "mul_scalar") else m * w
        r.intt_all_in_place() if hasattr(r, "intt_all_in_place") else r.ensure_coeff()
        mx.gpu_device_sync()
        t2 = time.perf_counter()
        ev = C.c_void_p()
        _ffi.check_status(lib.gpu_matrix_store_rns_batch(r.raw, buf_out, n * L * 8, 0, C.byref(ev)), "store")
        _ffi.wait_and_destroy_events(ev)
        t3 = time.perf_counter()
        best = min(best, t3 - t0)
        parts = (t1 - t0, t2 - t1, t3 - t2)
    print(f"{label}: total {best*1e3:.1f} ms  (load {parts[0]*1e3:.1f} = {nbytes/parts[0]/1e9:.1f} GB/s, step {parts[1]*1e3:.2f}, "
          f"store {parts[2]*1e3:.1f} = {nbytes/parts[2]/1e9:.1f} GB/s) -> {polys/best:.0f} ring-ops/s PCIe-inclusive")


out = np.empty_like(host)
run(host.ctypes.data, out.ctypes.data, "pageable")
pin_in, pin_out = lib.gpu_pinned_alloc(nbytes), lib.gpu_pinned_alloc(nbytes)
C.memmove(pin_in, host.ctypes.data, nbytes)
run(pin_in, pin_out, "pinned  ")
lib.gpu_pinned_free(pin_in); lib.gpu_pinned_free(pin_out)
