#!/bin/bash
mkdir -p gpurun_out
for e in fused scatter fused scatter; do for w in m3a m4; do MXX_HIP_SAMPLER_EXIT=$e timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$e $w', round(d['ms_per_step'],4), round(d['repeats']['median_ms_per_step'],4), d.get('kernel_launches_per_step'))"; done; done
