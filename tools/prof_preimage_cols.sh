#!/bin/bash
# kernel-trace stats of preimage calls with $1 target columns (what a rank sees under column sharding)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
COLS=${1:-7}
OUT=gpurun_out/pre_cols_$COLS
mkdir -p $OUT
cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import mxx_amd as mx
n, L = 16384, 10
p = mx.GpuDCRTPolyParams(n, mx.gen_crt_basis(n, L, 24), 12)
s = mx.GpuDCRTPolyTrapdoorSampler(p, 4.578)
td, pub = s.trapdoor(p, 1)
t = mx.GpuDCRTPolyUniformSampler().sample_uniform(p, 1, $COLS, mx.DistType.FinRingDist())
for _ in range(21):
    x = s.preimage(p, td, pub, t)
mx.gpu_device_sync()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/log.txt 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv && rm -rf $OUT/trace
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per call %.3f ms, launches per call %.1f" % (tot / 21e6, sum(int(r["Calls"]) for r in rows) / 21))
for r in rows[:14]:
    print("%-60s calls/call %5.1f avg %8.1f us  per call %7.3f ms" % (r["Name"][:60], int(r["Calls"]) / 21, float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 21e6))
PY
