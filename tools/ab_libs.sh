#!/bin/bash
# same-box A/B of library builds: tools/ab_libs.sh WORKLOAD lib1.so lib2.so ... (each run twice, interleaved)
WL=$1; shift
for round in 1 2; do for lib in "$@"; do
MXX_GPUPOLY_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline --repeats 2 >/dev/null 2>&1; python -c "
import json,sys
d=json.load(open('bench_detail.json')); k=d.get('kernels',{}); lb=k.get('large_batch',{})
print('$lib', 'step', round(d['ms_per_step'],4), ' '.join(f\"{n}={v['us']}\" for n,v in k.items() if isinstance(v,dict) and 'us' in v), '| large:', ' '.join(f\"{n}={v['ns_per_vector']}\" for n,v in lb.items() if isinstance(v,dict)))"
done; done
