"""Time individual ABI calls at BASELINE shapes (hipEvent marks on the engine stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx

n = 16384
def timed(ctx, fn, reps=3):
    fn()
    mx.gpu_device_sync()
    best = 1e9
    for _ in range(reps):
        ctx.timer_start(); r = fn(); ms = ctx.timer_stop(); best = min(best, ms)
    return best

which = sys.argv[1] if len(sys.argv) > 1 else "dec"
if which == "dec":
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 8, 24), 12)
    us = mx.GpuDCRTPolyUniformSampler()
    M = us.sample_uniform(p, 64, 64, mx.DistType.FinRingDist())
    ctx = p.ctx()
    k = p.modulus_digits()
    print("decompose 64x64 -> %dx64 (EVAL in, EVAL out): %.2f ms" % (64 * k, timed(ctx, lambda: M.decompose())))
    Mc = M.clone().into_coeff_domain()
    out = mx.GpuDCRTPolyMatrix(p, 64 * k, 64, 7, False)
    from mxx_amd import _ffi
    print("decompose kernel only (COEFF in, COEFF out): %.2f ms" % timed(ctx, lambda: _ffi.check_status(_ffi.lib().gpu_matrix_decompose_base(Mc.raw, 12, out.raw), "dec")))
    S = us.sample_uniform(p, 8, 64 * k, mx.DistType.FinRingDist())
    print("mul_decompose (8 x %d) * G^-1(64x64), chunk 1: %.2f ms" % (64 * k, timed(ctx, lambda: S.mul_decompose(M), 1)))
    os.environ["MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"] = "64"
    print("mul_decompose chunk 64: %.2f ms" % timed(ctx, lambda: S.mul_decompose(M), 2))
    out2 = mx.GpuDCRTPolyMatrix(p, 8, 64, 7, True)
    print("gpupoly_matrix_mul_decompose (C ABI, one call): %.2f ms" % timed(ctx, lambda: _ffi.check_status(_ffi.lib().gpupoly_matrix_mul_decompose(out2.raw, S.raw, M.raw, 12), "mul_decompose"), 2))
    D = M.decompose()
    print("product (8x%d)*(%dx64) alone: %.2f ms" % (64 * k, 64 * k, timed(ctx, lambda: S * D, 2)))
if which == "tensor":
    # S (1 x rows_b*k*ident) * (I_ident (x) G^-1(B)), B rows_b x cols_b: a BGG-style row-vector evaluation step
    p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, 8, 24), 12)
    us = mx.GpuDCRTPolyUniformSampler()
    ctx = p.ctx()
    k = p.modulus_digits()
    rows_b, cols_b, ident = 4, 16, 4
    B = us.sample_uniform(p, rows_b, cols_b, mx.DistType.FinRingDist())
    S = us.sample_uniform(p, 1, rows_b * k * ident, mx.DistType.FinRingDist())
    S0 = us.sample_uniform(p, 1, rows_b * ident, mx.DistType.FinRingDist())
    os.environ.pop("MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH", None)
    print("mul_tensor_identity_decompose (1 x %d) * (I_%d (x) G^-1(%dx%d)), extension: %.2f ms" % (S.ncol, ident, rows_b, cols_b, timed(ctx, lambda: S.mul_tensor_identity_decompose(B, ident))))
    print("mul_tensor_identity (1 x %d) * (I_%d (x) %dx%d), extension: %.3f ms" % (S0.ncol, ident, rows_b, cols_b, timed(ctx, lambda: S0.mul_tensor_identity(B, ident))))
    os.environ["MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"] = "1"
    print("mul_tensor_identity_decompose, the reference wrapper's loop: %.2f ms" % timed(ctx, lambda: S.mul_tensor_identity_decompose(B, ident), 2))
    print("mul_tensor_identity, the reference wrapper's loop: %.3f ms" % timed(ctx, lambda: S0.mul_tensor_identity(B, ident), 2))
    key = bytes(range(32))
    hs = mx.GpuDCRTPolyHashSampler()
    d = mx.DistType.FinRingDist()
    print("sample_hash_decomposed 8x8 (extension): %.2f ms" % timed(ctx, lambda: hs.sample_hash_decomposed(p, key, b"t", 8, 8, d)))
    print("sample_hash(...).decompose() 8x8 (two calls): %.2f ms" % timed(ctx, lambda: hs.sample_hash(p, key, b"t", 8, 8, d).decompose()))
    M = us.sample_uniform(p, 30, 120, d)
    print("transpose 30x120: %.3f ms" % timed(ctx, lambda: M.transpose()))
if which == "batch":
    # a "level" of 16 small products on the M4 ring (n = 256, 51-bit, L = 12): one call per product vs one batched call
    p = mx.GpuDCRTPolyParams(256, mx.gen_crt_basis(256, 12, 51), 17)
    us = mx.GpuDCRTPolyUniformSampler()
    ctx = p.ctx()
    d = mx.DistType.FinRingDist()
    ls = [us.sample_uniform(p, 1, 76, d) for _ in range(16)]
    rs = [us.sample_uniform(p, 76, 4, d) for _ in range(16)]
    print("16 x (1x76)*(76x4), n=256, L=12: one call each %.3f ms" % timed(ctx, lambda: [l * r for l, r in zip(ls, rs)], 5))
    print("16 x (1x76)*(76x4): gpupoly_matrix_mul_batch %.3f ms" % timed(ctx, lambda: mx.GpuDCRTPolyMatrix.mul_batch(ls, rs), 5))
