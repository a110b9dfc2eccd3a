#!/bin/bash
# round 3, first GPU pass: new tests, default bench, in-process rehearsal, sampler timing
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_comm.py tests/test_gpu_parallel.py -x -q > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03a_tests.log
tail -5 gpurun_out/r03a_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r03a_bench_default.json 2> gpurun_out/r03a_bench_default.err
echo "bench rc=$?"
MXX_BENCH_INPROC_SHARE_DEVICES=1 timeout -k 10 600 python bench.py --gpus 2 --inproc --steps 10 > gpurun_out/r03a_bench_inproc2.json 2> gpurun_out/r03a_bench_inproc2.err
echo "inproc rc=$?"
tail -3 gpurun_out/r03a_bench_inproc2.err
timeout -k 10 300 python tools/time_sampler.py > gpurun_out/r03a_time_sampler.txt 2>&1
echo "sampler rc=$?"
cat gpurun_out/r03a_time_sampler.txt
