#!/bin/bash
# usage: tools/kstats_color.sh variants/*.so -- per-kernel durations of the colour-gradient step per library build
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so)
  DEPTHWARP_LIB=$R/$so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kc_$n -- python3 $R/tools/time_color_grad.py > /dev/null 2>&1
  echo "== $n"
  f=$(find $R/gpurun_out/kc_$n -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'dw::' in r['Name'] or 'fillBuffer' in r['Name']:
        print(f"  {r['Name'][:60]:60s} {float(r['AverageNs'])/1000:8.1f} us")
PY
done
