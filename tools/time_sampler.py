"""Times the distribution samplers on the bench operands: sample_uniform of M2A's B (30 x 120, L = 15: 3.54 GB) and of a
64 x 64, L = 8 matrix, bit / ternary / Gaussian on the preimage's p2 shape; the sampler kernel and the forward transform
that follows it are reported separately (keep_coeff through gpupoly_matrix_sample_decomposed is not used: plain marks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

N = 16384


def seed(tag):
    return mx.GpuRngSeed.from_bytes(bytes([(tag * 37 + i * 11 + 5) & 0xFF for i in range(32)]))


def timed(ctx, fn, reps=5):
    fn()
    out = []
    for r in range(reps):
        ctx.timer_start()
        fn()
        out.append(ctx.timer_stop())
    return min(out), sorted(out)[len(out) // 2]


def main():
    lib = _ffi.lib()
    for name, depth, rows, cols in (("M2A B", 15, 30, 120), ("64x64 L=8", 8, 64, 64), ("p2 (20x50, L=10)", 10, 20, 50)):
        p = mx.GpuDCRTPolyParams(N, mx.gen_crt_basis(N, depth, 24), 12)
        ctx = p.ctx()
        m = mx.GpuDCRTPolyMatrix(p, rows, cols, depth - 1, True)
        gb = rows * cols * depth * N * 4 / 1e9
        for dist, dname, sigma in ((0, "uniform", 0.0), (2, "bit", 0.0), (3, "ternary", 0.0), (1, "gauss", 4.578)):
            if dist == 1 and rows * cols > 2000:
                continue
            full = lambda: _ffi.check_status(lib.gpu_matrix_sample_distribution(m.raw, dist, sigma, seed(3)), "sample")
            ntt = lambda: (_ffi.check_status(lib.gpu_matrix_intt_all(m.raw), "intt"), _ffi.check_status(lib.gpu_matrix_ntt_all(m.raw), "ntt"))
            mn, med = timed(ctx, full)
            nmn, _ = timed(ctx, ntt)
            print(f"{name:18s} {dname:8s}: sample+NTT {mn:8.3f} ms (median {med:.3f}); INTT+NTT pair {nmn:7.3f} ms -> sampler alone ~{mn - nmn / 2:7.3f} ms "
                  f"= {gb / max(mn - nmn / 2, 1e-6) * 1e3 / 1e3:6.2f} TB/s of output ({gb:.2f} GB)")
        del m


if __name__ == "__main__":
    main()
