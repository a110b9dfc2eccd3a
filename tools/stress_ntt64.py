"""One-off confidence run for the double-precision 64-bit-word transforms: many random and structured vectors, several
moduli per bit width, forward and inverse, against the integer kernels (MXX_HIP_NTT64=int) - no CPU in the loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mxx_amd as mx
from mxx_amd import _ffi

rng = np.random.default_rng(12345)
bad = 0
for logn in (10, 11, 12, 13, 14):
    n = 1 << logn
    for bits in (51, 50, 49, 47, 40):
        moduli = mx.gen_crt_basis(n, 6, bits)
        p = mx.GpuDCRTPolyParams(n, moduli, 17)
        q = np.asarray(moduli, dtype=np.uint64).reshape(1, 1, -1, 1)
        polys = 48 if logn <= 12 else 16
        x = (rng.integers(0, 1 << 62, size=(polys, 1, 6, n), dtype=np.uint64) % q).astype(np.uint64)
        x[0] = q - 1                      # all maximal
        x[1] = (q - 1) * (np.arange(n) % 2).astype(np.uint64)
        x[2] = (q - 1) * ((np.arange(n) // (n // 2)) % 2).astype(np.uint64)
        x[3] = np.where(rng.integers(0, 2, size=(1, 6, n)) == 1, q[0] - 1, 0)
        res = {}
        for mode in ("f64", "int"):
            if mode == "int":
                os.environ["MXX_HIP_NTT64"] = "int"
            else:
                os.environ.pop("MXX_HIP_NTT64", None)
            _ffi.reload_env()
            m = mx.GpuDCRTPolyMatrix.from_rns(p, x, False)
            m.ntt_all_in_place()
            ev = m.to_rns()
            e = mx.GpuDCRTPolyMatrix.from_rns(p, x, True)
            e.intt_all_in_place()
            res[mode] = (ev, e.to_rns())
            m.intt_all_in_place()
            assert np.array_equal(m.to_rns(), x), (logn, bits, mode, "round trip")
        ok = np.array_equal(res["f64"][0], res["int"][0]) and np.array_equal(res["f64"][1], res["int"][1])
        bad += 0 if ok else 1
        print(f"n=2^{logn} {bits}-bit x6 limbs, {polys} polys: {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
