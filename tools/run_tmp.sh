#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/time_ntt64.py 2>&1 | tee gpurun_out/r03_time_ntt64.txt
