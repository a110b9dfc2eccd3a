#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=gpurun_out/m4prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small -- python3 bench.py --workload m4 --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline > $OUT/small.log 2>&1
cp $OUT/small/*/*kernel_stats.csv $OUT/kernel_stats_small.csv
grep -i "ntt_" $OUT/kernel_stats_small.csv | cut -c1-60,100-260
export MXX_HIP_NTT_PATH=generic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/generic -- python3 bench.py --workload m4 --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline > $OUT/generic.log 2>&1
cp $OUT/generic/*/*kernel_stats.csv $OUT/kernel_stats_generic.csv
grep -i "ntt_" $OUT/kernel_stats_generic.csv | cut -c1-60,100-260
rm -rf $OUT/small $OUT/generic
