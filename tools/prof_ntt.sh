#!/bin/bash
# PMC passes for the M1 bench (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
run() { rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline ${@:3} > $OUT/$1.log 2>&1; }
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "${@:2}"
run sq2 "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "${@:2}"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "${@:2}"
run fetch "FETCH_SIZE" "${@:2}"
run write "WRITE_SIZE GRBM_GUI_ACTIVE" "${@:2}"
python3 tools/pmc_summary.py $OUT ntt > $OUT/summary_ntt.txt
python3 tools/pmc_summary.py $OUT "" > $OUT/summary_all.txt
cat $OUT/summary_ntt.txt
