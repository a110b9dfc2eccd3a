"""The 64-bit-word path (51-bit limbs) away from n = 256: algorithmic GB/s of the main entry points at n = 2^13, L = 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

d = mx.DistType.FinRingDist()
us = mx.GpuDCRTPolyUniformSampler()

def timed(ctx, fn, reps=3):
    fn(); mx.gpu_device_sync()
    best = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); ms = ctx.timer_stop(); best = min(best, ms)
    return best

for (n, L, bits, base) in ((8192, 4, 51, 17), (8192, 8, 24, 12)):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, bits), base)
    ctx = p.ctx()
    pb = L * n * ctx.word_bytes()
    k = p.modulus_digits()
    print(f"--- n={n} L={L} {bits}-bit words={ctx.word_bytes()} bytes, k={k}, poly {pb} B")
    a = us.sample_uniform(p, 64, 64, d); b = us.sample_uniform(p, 64, 64, d)
    print("add 64x64                 %8.3f ms %8.1f GB/s" % ((t := timed(ctx, lambda: a + b)), 3 * 4096 * pb / t / 1e6))
    print("mul_scalar 64x64          %8.3f ms %8.1f GB/s" % ((t := timed(ctx, lambda: a.mul_scalar(b.slice(0, 1, 0, 1)))), 2 * 4096 * pb / t / 1e6))
    print("ntt+intt 64x64            %8.3f ms %8.1f GB/s" % ((t := timed(ctx, lambda: a.clone().into_coeff_domain())), 4 * 4096 * pb / t / 1e6))
    for (r, kk, c) in ((1, 30, 120), (64, 64, 64), (8, 256, 64), (2, 72, 4)):
        x = us.sample_uniform(p, r, kk, d); y = us.sample_uniform(p, kk, c, d)
        t = timed(ctx, lambda: x * y)
        print("matmul (%dx%d)(%dx%d)   %8.3f ms %8.1f GB/s algorithmic, %6.2f ns per ring-MAC" % (r, kk, kk, c, t, (r * kk + kk * c + r * c) * pb / t / 1e6, t * 1e6 / (r * kk * c)))
    s = us.sample_uniform(p, 8, 8, d)
    t = timed(ctx, lambda: s.decompose())
    print("decompose 8x8 -> %dx8     %8.3f ms %8.1f GB/s" % (8 * k, t, (64 + 64 * k) * pb / t / 1e6))
    t = timed(ctx, lambda: us.sample_uniform(p, 8, 8, mx.DistType.GaussDist(4.578)))
    print("gauss 8x8                 %8.3f ms %8.1f Msamples/s" % (t, 64 * n / t / 1e3))
    samp = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
    td, pub = samp.trapdoor(p, 1)
    tg = us.sample_uniform(p, 1, 8, d)
    t = timed(ctx, lambda: samp.preimage(p, td, pub, tg), 2)
    print("preimage 8 columns        %8.3f ms" % t)
    cb = None
    t = timed(ctx, lambda: a.slice(0, 8, 0, 8).to_compact_bytes(), 2)
    print("to_compact_bytes 8x8      %8.3f ms" % t)
    del a, b, s
