#!/bin/bash
# SQ issue/stall counters for one bench workload (runs on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-m1}
OUT=gpurun_out/pmc_sq_$WL
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
run b "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"
run c "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
python3 tools/pmc_summary.py $OUT "${2:-ntt14}" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
tail -3 $OUT/a.log $OUT/b.log $OUT/c.log
