#!/bin/bash
# full GPU suite + the bench lines
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03c_tests.log
tail -6 gpurun_out/r03c_tests.log
