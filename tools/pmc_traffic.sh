#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels of one bench workload matching $2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-m3a}; SUB=${2:-}
OUT=gpurun_out/pmc_traffic_$WL
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/w.log 2>&1
python3 tools/pmc_summary.py $OUT "$SUB"
