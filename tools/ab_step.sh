#!/bin/bash
# same-box A/B of library builds on ms_per_step only: tools/ab_step.sh "WL1 WL2" lib1.so lib2.so ... (two interleaved rounds)
WLS=$1; shift
for round in 1 2; do for wl in $WLS; do for lib in "$@"; do
MXX_GPUPOLY_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --repeats 3 >/dev/null 2>&1; python -c "
import json,sys
d=json.load(open('bench_detail.json'))
print('$wl', '$lib', 'ms_per_step', round(d['ms_per_step'],4))"
done; done; done
