#!/bin/bash
# instruction-mix counters for the kernels of one bench workload whose name contains $2 (GPU box, via gpurun)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-m3a}; SUB=${2:-}
OUT=gpurun_out/pmc_kernel_$WL
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/b -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
python3 tools/pmc_summary.py $OUT "$SUB" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
