#!/bin/bash
# usage: tools/ab.sh variants/*.so  -- time the dominant kernel for each library build (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for so in "$@"; do
  DEPTHWARP_LIB=$R/$so timeout -k 10 120 python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', 'step_ms', round(d['ms_per_step'],4), 'tile_ms', round(d['roofline']['kernel_ms'],4), 'loss', d['loss'])" || echo "$so FAILED"
done
