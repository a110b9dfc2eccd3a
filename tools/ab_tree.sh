#!/bin/bash
# same-box A/B of two trees: the repo and its _old worktree
for round in 1 2; do for wl in m3a m3b m4; do for t in _old .; do
(cd $GRAFT_REPO_ROOT/$t && timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl', '$t', 'ms_per_step', round(d['ms_per_step'],4))")
done; done; done
