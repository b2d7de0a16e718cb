#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for so in "$@"; do
  DEPTHWARP_LIB=$R/$so timeout -k 10 120 python3 $R/bench_next.py --only soft_argmin --steps 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$so', d['metric'].split()[0], round(d['ms_per_step'],4), 'ms', round(d['roofline']['achieved']), 'GB/s')"
done
