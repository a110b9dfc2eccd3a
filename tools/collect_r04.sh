#!/bin/bash
# Round-4 evidence, run on the GPU box through gpurun from the repo root.  Stages (STAGES="a b c", default all):
#   a  the default bench line (all BASELINE configs) + the per-workload lines
#   b  rocprofv3 --pmc passes per workload (FETCH_SIZE, WRITE_SIZE, SQ counters: each in its own run, never combined with
#      other trace domains), reduced to profiles-ready JSON by tools/pmc_window.py (only the launches between bench.py's
#      region markers are counted)
#   c  rocprofv3 --kernel-trace --stats of the default bench
# Results: gpurun_out/r04/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
STAGES=${STAGES:-a b c}
for S in $STAGES; do
case $S in
a)
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { echo "bench default failed"; tail -20 $OUT/bench_default.err; exit 1; }
  tail -c 400 $OUT/bench_default.json; echo
  ;;
b)
  for WL in ${PMC_WORKLOADS:-m3a m3b m4 m2b_decompose m2b_mul_decompose m2b m2a m1}; do
    STEPS=3
    CMD="python3 bench.py --workload $WL --steps $STEPS --warmup 1 --repeats 0 --no-cpu-baseline --no-trace"
    for PASS in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
      TAG=${PASS%%:*}; CTRS=${PASS#*:}
      rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_${WL}_$TAG -- $CMD > $OUT/pmc_${WL}_$TAG.log 2>&1 || { echo "pmc $WL $TAG failed"; tail -5 $OUT/pmc_${WL}_$TAG.log; exit 1; }
    done
    python3 tools/pmc_window.py --workload $WL --steps $STEPS --out $OUT/r04_pmc_$WL.json $OUT/pmc_${WL}_fetch $OUT/pmc_${WL}_write $OUT/pmc_${WL}_sq > $OUT/r04_pmc_$WL.txt || exit 1
    rm -rf $OUT/pmc_${WL}_fetch $OUT/pmc_${WL}_write $OUT/pmc_${WL}_sq
    echo "== $WL"; head -12 $OUT/r04_pmc_$WL.txt
  done
  ;;
c)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 bench.py --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline --no-trace > $OUT/under_rocprof_default.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/under_rocprof_default.log; exit 1; }
  cp $OUT/trace_default/*/*kernel_stats.csv $OUT/r04_kernel_stats_default.csv 2>/dev/null
  rm -rf $OUT/trace_default
  head -12 $OUT/r04_kernel_stats_default.csv | cut -c1-160
  ;;
esac
done
