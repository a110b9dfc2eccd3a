#!/bin/bash
# per-kernel averages of one workload under two library builds: tools/prof_ab.sh WL lib1.so lib2.so
WL=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export MXX_GPUPOLY_LIB=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/prof_$lib
  (cd $GRAFT_REPO_ROOT && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o out --output-format csv -- python3 bench.py --workload $WL --no-cpu-baseline --repeats 1 > /dev/null 2>&1)
  f=$(find /tmp/prof_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
done
