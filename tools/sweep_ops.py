"""Algorithmic GB/s of the data-movement and elementwise entry points at several shapes: anything far below the
streaming rates (5-6.8 TB/s for large operands) is a pathology to look at.  n = 2^14 unless noted."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n = 16384
d = mx.DistType.FinRingDist()
us = mx.GpuDCRTPolyUniformSampler()

def timed(ctx, fn, reps=3):
    fn(); mx.gpu_device_sync()
    best = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); ms = ctx.timer_stop(); best = min(best, ms)
    return best

def report(name, ms, nbytes):
    print(f"{name:58s} {ms:8.3f} ms  {nbytes / ms / 1e6:8.1f} GB/s", flush=True)

for L in (4, 10):
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
    ctx = p.ctx()
    pb = L * n * 4
    for (r, c) in ((1, 1), (1, 50), (22, 50), (64, 64)):
        a = us.sample_uniform(p, r, c, d)
        b = us.sample_uniform(p, r, c, d)
        s = us.sample_uniform(p, 1, 1, d)
        tag = f"L={L} {r}x{c}"
        report(f"{tag} add", timed(ctx, lambda: a + b), 3 * r * c * pb)
        report(f"{tag} mul_scalar", timed(ctx, lambda: a.mul_scalar(s)), 2 * r * c * pb)
        report(f"{tag} neg", timed(ctx, lambda: -a), 2 * r * c * pb)
        report(f"{tag} clone", timed(ctx, lambda: a.clone()), 2 * r * c * pb)
        report(f"{tag} transpose", timed(ctx, lambda: a.transpose()), 2 * r * c * pb)
        report(f"{tag} ntt+intt (clone, 2 transforms)", timed(ctx, lambda: a.clone().into_coeff_domain()), 4 * r * c * pb)
        if r >= 2 and c >= 2:
            report(f"{tag} slice (half rows, half cols)", timed(ctx, lambda: a.slice(0, r // 2, 0, c // 2)), 2 * (r // 2) * (c // 2) * pb)
            report(f"{tag} slice_columns(1 col)", timed(ctx, lambda: a.slice_columns(0, 1)), 2 * r * pb)
            report(f"{tag} concat_columns", timed(ctx, lambda: a.concat_columns([b])), 4 * r * c * pb)
            report(f"{tag} concat_rows", timed(ctx, lambda: a.concat_rows([b])), 4 * r * c * pb)
            report(f"{tag} concat_diag", timed(ctx, lambda: a.concat_diag([b])), 8 * r * c * pb)
            report(f"{tag} vectorize_columns", timed(ctx, lambda: a.vectorize_columns()), 2 * r * c * pb)
        report(f"{tag} zero", timed(ctx, lambda: mx.GpuDCRTPolyMatrix.zero(p, r, c)), r * c * pb)
        report(f"{tag} equal", timed(ctx, lambda: a == b), 2 * r * c * pb)
        if r == c and r <= 22:
            report(f"{tag} identity", timed(ctx, lambda: mx.GpuDCRTPolyMatrix.identity(p, r)), r * c * pb)
            report(f"{tag} gadget_matrix", timed(ctx, lambda: mx.GpuDCRTPolyMatrix.gadget_matrix(p, r)), r * r * p.modulus_digits() * pb)
        if r * c <= 64:
            k = p.modulus_digits()
            report(f"{tag} decompose -> {r * k}x{c}", timed(ctx, lambda: a.decompose()), (r * c + r * k * c) * pb)
        report(f"{tag} sample_uniform", timed(ctx, lambda: us.sample_uniform(p, r, c, d)), r * c * pb)
        report(f"{tag} sample bit", timed(ctx, lambda: us.sample_uniform(p, r, c, mx.DistType.BitDist())), r * c * pb)
        del a, b, s
