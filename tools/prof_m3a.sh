#!/bin/bash
# kernel-trace stats + issue counters of one preimage bench (M3A); results in gpurun_out/m3a/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/m3a
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload m3a --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline > $OUT/under_rocprof.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats_m3a.csv && rm -rf $OUT/trace
python3 tools/kstats.py $OUT 14
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --workload m3a --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
python3 tools/pmc_summary.py $OUT "" > $OUT/pmc_summary_m3a.txt 2>&1
rm -rf $OUT/pmc_sq
head -40 $OUT/pmc_summary_m3a.txt
