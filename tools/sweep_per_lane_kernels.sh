#!/bin/bash
# per-kernel times of an M3A call against MXX_HIP_SAMPLER_PER_LANE (0 = the launcher's rule), two interleaved rounds
for round in 1 2; do for pl in 0 1 2 3 4 6 8; do
if [ $pl = 0 ]; then unset MXX_HIP_SAMPLER_PER_LANE; else export MXX_HIP_SAMPLER_PER_LANE=$pl; fi
timeout -k 10 300 python bench.py --workload m3a --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d.get('roofline',{})
ks=' '.join(f\"{k['kernel'].split('::')[-1][:22]}={k['ms']:.3f}\" for k in r.get('kernels',[])[:7] if 'lanes' in k['kernel'] or 'gauss' in k['kernel'])
print('per_lane=$pl', 'ms_per_step', round(d['repeats']['median_ms_per_step'],4), '|', ks)"
done; done
