"""Compact wire format (to_compact_bytes / from_compact_bytes) timings: Gaussian-sized and uniform matrices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n = 16384
for L, rows, cols, dist, label in ((10, 22, 50, mx.DistType.GaussDist(1.5e6), "Gaussian sigma=1.5e6 (preimage-sized values)"),
                                   (10, 8, 8, mx.DistType.FinRingDist(), "uniform mod Q"),
                                   (15, 1, 30, mx.DistType.FinRingDist(), "uniform mod Q")):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
    m = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, rows, cols, dist)
    m.intt_all_in_place()
    mx.gpu_device_sync()
    blob = m.to_compact_bytes()
    best_s = best_l = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); blob = m.to_compact_bytes(); best_s = min(best_s, time.perf_counter() - t0)
        t0 = time.perf_counter(); back = mx.GpuDCRTPolyMatrix.from_compact_bytes(p, blob); mx.gpu_device_sync(); best_l = min(best_l, time.perf_counter() - t0)
    assert back == m
    coeffs = rows * cols * n
    print(f"{rows}x{cols} L={L} {label}: {len(blob)/1e6:.1f} MB payload; store {best_s*1e3:.2f} ms ({coeffs/best_s/1e9:.2f} G coeff/s), load {best_l*1e3:.2f} ms")

# where the store's time goes for the first shape: the ABI call alone vs the host framing
import ctypes as C
from mxx_amd import _ffi
L, rows, cols = 10, 22, 50
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
m = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, rows, cols, mx.DistType.GaussDist(1.5e6))
m.intt_all_in_place()
cap = rows * cols * n * 240 // 8
for kind in ("bytearray", "pinned"):
    if kind == "bytearray":
        backing = bytearray(cap); buf = (C.c_uint8 * cap).from_buffer(backing)
    else:
        _ffi.lib().gpu_pinned_alloc.restype = C.c_void_p
        ptr = _ffi.lib().gpu_pinned_alloc(cap); buf = C.cast(ptr, C.POINTER(C.c_uint8))
    bits, bpc, plen = C.c_uint16(0), C.c_uint16(0), C.c_size_t(0)
    for rep in range(2):
        t0 = time.perf_counter()
        _ffi.check_status(_ffi.lib().gpu_matrix_store_compact_bytes(m.raw, buf, cap, C.byref(bits), C.byref(bpc), C.byref(plen)), "store")
        dt = time.perf_counter() - t0
    print(f"ABI store into {kind} host memory: {dt*1e3:.2f} ms for {plen.value/1e6:.1f} MB")

# RNS staging bytes (to_cpu_staging_bytes / from_cpu_staging_bytes, gpu_dcrt_poly.rs:1046-1079): u64 wire layout
for L, rows, cols in ((10, 8, 8), (10, 22, 50)):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
    m = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, rows, cols, mx.DistType.FinRingDist())
    mx.gpu_device_sync()
    bs = bl = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); blob = m.to_cpu_staging_bytes(); bs = min(bs, time.perf_counter() - t0)
        t0 = time.perf_counter(); back = mx.GpuDCRTPolyMatrix.from_cpu_staging_bytes(p, blob); mx.gpu_device_sync(); bl = min(bl, time.perf_counter() - t0)
    assert back == m
    print(f"staging bytes {rows}x{cols} L={L}: {len(blob)/1e6:.0f} MB; store {bs*1e3:.1f} ms ({len(blob)/bs/1e9:.1f} GB/s), load {bl*1e3:.1f} ms ({len(blob)/bl/1e9:.1f} GB/s)")
