#!/bin/bash
# usage: tools/kstats.sh variants/*.so -- per-kernel average durations of a short bench run per library build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so)
  DEPTHWARP_LIB=$R/$so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$n -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  echo "== $n"
  f=$(find $R/gpurun_out/ks_$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if 'dw::' in r['Name']:
        print(f"  {r['Name'][:60]:60s} {float(r['AverageNs'])/1000:8.1f} us"); tot += float(r['AverageNs']) / 1000
print(f"  {'sum':60s} {tot:8.1f} us")
PY
done
