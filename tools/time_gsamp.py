"""Times the G-lattice sampler (gpu_matrix_gauss_samp_gq_arb_base) and the p1 sampler on bulk inputs at 1..4 digits per
tower; run once per MXX_HIP_SAMPLER_FILL_EVERY setting (the switch is read when the context is created):
    for k in 1 2 3 4; do MXX_HIP_SAMPLER_FILL_EVERY=$k python tools/time_gsamp.py; done"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mxx_amd as mx
from mxx_amd import _ffi

N = 16384


def seed(tag):
    return mx.GpuRngSeed.from_bytes(bytes([(tag * 37 + i * 11 + 5) & 0xFF for i in range(32)]))


def main():
    lib = _ffi.lib()
    every = os.environ.get("MXX_HIP_SAMPLER_FILL_EVERY", "default")
    # (limb bits, base bits) -> digits per tower
    for bits, base, depth, cols in ((24, 24, 8, 8), (24, 12, 8, 8), (51, 17, 4, 8), (24, 6, 4, 8)):
        p = mx.GpuDCRTPolyParams(N, mx.gen_crt_basis(N, depth, bits), base)
        ctx = p.ctx()
        dpt = -(-bits // base)
        src = mx.GpuDCRTPolyMatrix.sample_distribution(p, 1, cols, 0, 0.0, seed(1))
        src.intt_all_in_place()
        out = mx.GpuDCRTPolyMatrix.new_empty(p, p.modulus_digits(), cols)
        c = (2 ** base + 1) * 4.578
        call = lambda: _ffi.check_status(lib.gpu_matrix_gauss_samp_gq_arb_base(src.raw, base, c, 4.578, seed(2), out.raw), "gsamp")
        call()
        ts = []
        for _ in range(5):
            ctx.timer_start()
            call()
            ts.append(ctx.timer_stop())
        elems = cols * depth * N
        print(f"fill_every={every} dpt={dpt} ({bits}-bit limbs, base 2^{base}, {elems / 1e6:.2f} M elements): "
              f"{min(ts):.3f} ms min, {sorted(ts)[2]:.3f} median, {min(ts) * 1e6 / (elems * dpt):.2f} ns per integer", flush=True)


if __name__ == "__main__":
    main()
