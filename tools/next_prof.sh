#!/bin/bash
# usage (GPU box): tools/next_prof.sh TAG WHICH -- one bench_next.py line under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o np -f csv -- python3 $R/bench_next.py --only $2 --no-cpu-baseline --steps 20 > $O/bench.txt 2>&1
python3 - $O <<'PY'
import csv, glob, json, sys
for l in open(sys.argv[1] + "/bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["metric"], round(d["ms_per_step"] * 1e3, 1), "us", round(d["roofline"]["frac"], 3))
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"][:50], r["Calls"], r["AverageNs"], r["Percentage"])
PY
