#!/bin/bash
# usage: tools/ab_env.sh LIB [VAR=VALUE ...] -- the bench line of one library build under extra environment (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=$1; shift
env "$@" DEPTHWARP_LIB=$R/$LIB timeout -k 10 120 python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$LIB $*', 'step_ms', round(d['ms_per_step'],4), d['config']['launch'], 'loss', d['loss'])" || echo "$LIB FAILED"
