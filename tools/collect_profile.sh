#!/bin/bash
# Runs on the GPU box (via gpurun) from the repo root: kernel-trace stats + PMC passes of the
# default bench; results land in gpurun_out/profile_$1/ and are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; WL=${2:-m1}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats_$WL.csv 2>/dev/null
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
python3 tools/pmc_summary.py $OUT "" > $OUT/pmc_summary_$WL.txt
python3 bench.py --workload $WL --steps 20 --warmup 3 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
tail -1 $OUT/bench_$WL.json | cut -c1-400
rm -rf $OUT/trace/*/*kernel_trace.csv $OUT/fetch $OUT/write $OUT/sq
