#!/bin/bash
# kernel-trace stats of preimage calls on the reference's end-to-end ring (n = 2^16, 28-bit limbs, base 2^14)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stats_pre16
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_preimage16.py > $OUT/log.txt 2>&1
cp $OUT/trace/*/*kernel_stats.csv gpurun_out/stats_pre16.csv && rm -rf $OUT/trace
cat $OUT/log.txt | tail -5
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/stats_pre16.csv")))
for r in rows[:22]:
    print("%-70s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
