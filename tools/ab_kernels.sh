#!/bin/bash
# same-box A/B of library builds with the per-kernel split of the launch trace:
#   tools/ab_kernels.sh "m3a m4" lib1.so lib2.so ...   (two interleaved rounds; prints ms per step and the top kernels' traced ms)
WLS=$1; shift
for round in 1 2; do for wl in $WLS; do for lib in "$@"; do
MXX_GPUPOLY_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --repeats 3 >/dev/null 2>&1; python -c "
import json,sys
d=json.load(open('bench_detail.json'))
r=d.get('roofline',{})
ks=' '.join(f\"{k['kernel'].split('::')[-1][:22]}={k['ms']:.3f}\" for k in r.get('kernels',[])[:6])
print('$wl', '$lib', 'ms_per_step', round(d['repeats']['median_ms_per_step'],4), '|', ks)"
done; done; done
