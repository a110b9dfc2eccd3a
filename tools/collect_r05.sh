#!/bin/bash
# Round-5 evidence, run on the GPU box through gpurun from the repo root:
#   MXX_HEAD=$(git rev-parse --short HEAD) gpurun --timeout 1200 -- "MXX_HEAD=$MXX_HEAD STAGES='a b c' bash tools/collect_r05.sh"
# Stages (STAGES="a b c", default all):
#   a  the default bench: the SHORT line the driver parses (bench_short.json) and the full record (bench_detail.json)
#   b  rocprofv3 --pmc passes per workload - FETCH_SIZE, WRITE_SIZE, the SQ issue counters, the lane counters - each in its
#      OWN run with no trace domain next to it (no --kernel-trace: tools/pmc_window.py reads counter_collection.csv only),
#      reduced to profiles-ready, VERSIONED records pmc_<workload>.json (commit, gpupoly_version, per-kernel ISA hash)
#   c  rocprofv3 --kernel-trace --stats of the default bench
# Results: gpurun_out/r05/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05
mkdir -p $OUT
STAGES=${STAGES:-a b c}
export MXX_HEAD=${MXX_HEAD:-unknown}
for S in $STAGES; do
case $S in
a)
  python3 bench.py > $OUT/bench_short.json 2> $OUT/bench_default.err || { echo "bench default failed"; tail -20 $OUT/bench_default.err; exit 1; }
  cp bench_detail.json $OUT/bench_detail.json
  wc -c $OUT/bench_short.json; cat $OUT/bench_short.json
  ;;
b)
  for WL in ${PMC_WORKLOADS:-m3a m3b m4 m4_batched m2b_decompose m2b_mul_decompose m2b m2a m1}; do
    STEPS=3
    CMD="python3 bench.py --workload $WL --steps $STEPS --warmup 1 --repeats 0 --sustain 0 --no-cpu-baseline --no-trace"
    for PASS in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "lanes:SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
      TAG=${PASS%%:*}; CTRS=${PASS#*:}
      rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_${WL}_$TAG -- $CMD > $OUT/pmc_${WL}_$TAG.log 2>&1 || { echo "pmc $WL $TAG failed"; tail -5 $OUT/pmc_${WL}_$TAG.log; exit 1; }
    done
    python3 tools/pmc_window.py --workload $WL --steps $STEPS --head $MXX_HEAD --out $OUT/pmc_$WL.json $OUT/pmc_${WL}_fetch $OUT/pmc_${WL}_write $OUT/pmc_${WL}_sq $OUT/pmc_${WL}_lanes > $OUT/pmc_$WL.txt || exit 1
    rm -rf $OUT/pmc_${WL}_fetch $OUT/pmc_${WL}_write $OUT/pmc_${WL}_sq $OUT/pmc_${WL}_lanes
    echo "== $WL"; head -12 $OUT/pmc_$WL.txt
  done
  ;;
c)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 bench.py --steps 20 --warmup 3 --repeats 0 --sustain 0 --no-cpu-baseline --no-trace > $OUT/under_rocprof_default.log 2>&1 || { echo "kernel trace failed"; tail -5 $OUT/under_rocprof_default.log; exit 1; }
  cp $OUT/trace_default/*/*kernel_stats.csv $OUT/r05_kernel_stats_default.csv 2>/dev/null
  rm -rf $OUT/trace_default
  head -12 $OUT/r05_kernel_stats_default.csv | cut -c1-160
  ;;
esac
done
