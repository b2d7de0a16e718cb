#!/bin/bash
# usage: tools/kstats.sh variants/*.so -- per-kernel average durations of a short bench run per library build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so)
  DEPTHWARP_LIB=$R/$so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$n -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  echo "== $n"; find $R/gpurun_out/ks_$n -name "*kernel_stats.csv" | head -1 | xargs cat | grep "dw::" | awk -F, '{gsub(/"/,"",$1); printf "  %-60s %8.1f us\n", substr($1,1,60), $4/1000}'
done
