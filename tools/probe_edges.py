"""One-off probes at sizes the unit tests do not reach (run on the GPU box); anything that fails becomes a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mxx_amd as mx

def check(name, fn):
    t = time.time()
    try:
        ok = fn()
        print(f"{name}: {'ok' if ok else 'FAIL'} ({time.time()-t:.1f}s)", flush=True)
    except Exception as e:
        print(f"{name}: EXC {e!r}", flush=True)

n = 16384
us = mx.GpuDCRTPolyUniformSampler()
U = mx.DistType.FinRingDist()

def compact_big():
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 8, 24), 12)
    m = us.sample_uniform(p, 32, 32, U)
    b = m.to_compact_bytes()
    return mx.GpuDCRTPolyMatrix.from_compact_bytes(p, b) == m
check("compact bytes 32x32, L=8, n=2^14", compact_big)

def preimage_d(d, depth, bits, base, cols):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, depth, bits), base)
    s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, A = s.trapdoor(p, d)
    t = us.sample_uniform(p, d, cols, U)
    x = s.preimage(p, td, A, t)
    return A * x == t
check("preimage d=2, L=4, 24-bit, 7 cols (padding)", lambda: preimage_d(2, 4, 24, 12, 7))
check("preimage d=3, L=3, 24-bit, 4 cols (m=6: old p1 kernel)", lambda: preimage_d(3, 3, 24, 12, 4))
check("preimage d=1, L=2, 51-bit (u64), 5 cols", lambda: preimage_d(1, 2, 51, 17, 5))
check("preimage d=2, L=2, 51-bit (u64), 3 cols", lambda: preimage_d(2, 2, 51, 17, 3))

def u64_ops():
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 3, 51), 17)
    a = us.sample_uniform(p, 9, 12, U); b = us.sample_uniform(p, 12, 10, U); c = us.sample_uniform(p, 12, 10, U)
    ok = a * (b + c) == a * b + a * c
    d = b.decompose()
    ok = ok and mx.GpuDCRTPolyMatrix.gadget_matrix(p, 12) * d == b
    ok = ok and mx.GpuDCRTPolyMatrix.from_compact_bytes(p, b.to_compact_bytes()) == b
    return ok
check("u64 words at n=2^14: product, decompose, compact bytes", u64_ops)

def big_decompose_l15():
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 15, 24), 12)
    m = us.sample_uniform(p, 8, 16, U)
    d = m.decompose()   # 240 x 16 polys x 15 limbs
    return mx.GpuDCRTPolyMatrix.gadget_matrix(p, 8) * d == m
check("decompose 8x16, L=15 (k=30)", big_decompose_l15)

def base_variants():
    ok = True
    for base in (1, 5, 8, 24):
        p = mx.GpuDCRTPolyParams(4096, mx.gen_crt_basis(4096, 2, 24), base)
        m = us.sample_uniform(p, 2, 2, U)
        ok = ok and mx.GpuDCRTPolyMatrix.gadget_matrix(p, 2) * m.decompose() == m
        if base >= 5:
            s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
            td, A = s.trapdoor(p, 1)
            t = us.sample_uniform(p, 1, 2, U)
            ok = ok and (A * s.preimage(p, td, A, t) == t or base == 24)
    return ok
check("base_bits 1/5/8/24 at n=4096", base_variants)
