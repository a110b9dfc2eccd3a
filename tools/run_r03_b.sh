#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_surface.py -x -q -k "gate_batch or mul_batch" > gpurun_out/r03b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03b_tests.log
tail -25 gpurun_out/r03b_tests.log
