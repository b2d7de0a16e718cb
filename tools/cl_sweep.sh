#!/bin/bash
# usage (GPU box): tools/cl_sweep.sh LIB WAVES... -- compute_loss bench lines for several launch sizes (LIB built with -DDW_TUNING)
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$1; shift
for w in "$@"; do
  echo "== $LIB waves $w"
  DEPTHWARP_LIB=$R/$LIB DW_CL_WAVES=$w timeout -k 10 200 python3 $R/bench_next.py --only compute_loss --no-cpu-baseline --steps 100 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ', d['config']['workload'][16:40], round(d['ms_per_step'] * 1e3, 1), 'us', round(d['roofline']['frac'], 3))
" || exit 1
done
