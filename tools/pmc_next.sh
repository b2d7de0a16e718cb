#!/bin/bash
# usage: tools/pmc_next.sh OUTDIR -- HBM traffic counters of the soft-argmin and cost-volume kernels (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
cd /tmp; export TMPDIR=/tmp
mkdir -p $OUT
for k in soft_argmin cost_volume; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $OUT/${k}_$c -- python3 $R/bench_next.py --only $k --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${k}_$c.log 2>&1
  done
done
python3 $R/tools/pmc_summary.py $OUT --json $OUT/summary.json
