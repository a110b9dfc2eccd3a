#!/bin/bash
# Phase timing of the forward 2^14 transform (mxx_amd/libgpupoly_phase.so = `tools/build_variant.sh phase PHASE_TIMING=1`):
# MXX_HIP_NTT_PHASE carries the phase mask there - 1: no global loads, 2: no butterflies, 4: no global stores,
# 8: coalesced stores (one more LDS trip), 16: cacheable instead of non-temporal stores for batches of 1 GiB and more.
#   gpurun -- bash tools/ab_ntt_phases.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
OUT=gpurun_out/r05; mkdir -p $OUT
: > $OUT/ntt_phases.txt
for MASK in ${MASKS:-0 8 16 24 3 11 19 27 0}; do
  MXX_GPUPOLY_LIB=$PWD/mxx_amd/libgpupoly_phase.so MXX_HIP_NTT_PHASE=$MASK python3 - "$MASK" >> $OUT/ntt_phases.txt 2> $OUT/ntt_phases.err <<'PY' || { tail -5 $OUT/ntt_phases.err; exit 1; }
import sys, statistics
import mxx_amd as mx
from mxx_amd import _ffi
lib = _ffi.lib()
N = 16384
names = {101: "butterflies only, twiddles from registers (no memory operation at all)", 64: "whole kernel, twiddles from registers", 37: "butterflies only: no loads, stores, LDS traffic or barriers", 32: "no LDS traffic / barriers", 8: "whole, coalesced stores", 16: "whole, cacheable stores", 24: "whole, coalesced + cacheable", 11: "stores only, coalesced", 19: "stores only, cacheable", 27: "stores only, coalesced + cacheable", 0: "whole kernel", 1: "no loads", 2: "no butterflies", 4: "no stores", 3: "stores only (+LDS)", 5: "butterflies only (+LDS)", 6: "loads only (+LDS)", 7: "LDS traffic + barriers only"}
for polys in (1024, 4096):
    p = mx.GpuDCRTPolyParams(N, mx.gen_crt_basis(N, 4, 24), 12)
    ctx = p.ctx()
    x = mx.GpuDCRTPolyMatrix.sample_distribution(p, polys, 1, mx.DistType.FinRingDist().as_ffi(), 0.0, mx.GpuRngSeed.from_bytes(bytes(range(32))))
    x.intt_all_in_place()
    ts = []
    for rep in range(12):
        ctx.timer_mark(100)
        _ffi.check_status(lib.gpu_matrix_ntt_all(x.raw), "ntt")
        ctx.timer_mark(101)
        x.is_ntt = False
        _ffi.check_status(lib.gpupoly_matrix_fill_zero(x.raw), "zero") if False else None
        mx.gpu_device_sync()
        if rep >= 2:
            ts.append(ctx.timer_elapsed(100, 101) * 1e3)
        # back to COEFF tag without transforming (timing only): the wrapper's tag is what gates the next call
        _ffi.check_status(lib.gpu_matrix_intt_all(x.raw), "intt")
    mask = int(sys.argv[1])
    print(f"mask {mask} ({names[mask]}), {polys} polys x 4 limbs: forward median {statistics.median(ts):.1f} us, min {min(ts):.1f} us, {statistics.median(ts) * 1e3 / (polys * 4):.2f} ns per vector")
    del x
PY
done
cat $OUT/ntt_phases.txt
