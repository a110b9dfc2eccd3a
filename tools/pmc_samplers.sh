#!/bin/bash
# issue counters of the sampler kernels on M3A (one rocprofv3 --pmc pass, kernel trace only): instructions per wave and
# per cycle of gauss_samp_lanes_kernel, sample_gauss_kernel, p1_sample_lanes_kernel.  Results: gpurun_out/pmc_samplers.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_samplers
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT -- python3 bench.py --workload m3a --steps 3 --warmup 1 --repeats 0 --no-cpu-baseline > $OUT/run.log 2>&1
python3 tools/pmc_summary.py $OUT lanes > gpurun_out/pmc_samplers.txt
python3 tools/pmc_summary.py $OUT sample_gauss >> gpurun_out/pmc_samplers.txt
rm -rf $OUT
cat gpurun_out/pmc_samplers.txt
