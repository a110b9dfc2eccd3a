#!/bin/bash
# kernel-trace stats of one bench workload: prof_stats.sh WORKLOAD [rows]; results in gpurun_out/stats_WORKLOAD.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-m3a}
OUT=gpurun_out/stats_$WL
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline > $OUT/under_rocprof.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv gpurun_out/stats_$WL.csv && rm -rf $OUT/trace
python3 - "$WL" "${2:-16}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("gpurun_out/stats_%s.csv" % sys.argv[1])))
for r in rows[: int(sys.argv[2])]:
    print("%-62s calls %4s avg %9.1f us total %8.2f ms" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
