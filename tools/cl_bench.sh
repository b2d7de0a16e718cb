#!/bin/bash
# usage (GPU box): tools/cl_bench.sh TAG -- compute_loss bench lines + rocprofv3 kernel stats into gpurun_out/TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
timeout -k 10 300 python3 $R/bench_next.py --only compute_loss ${CL_ARGS} > $O/bench.txt 2>&1 || { tail -5 $O/bench.txt; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof -o cl -f csv -- python3 $R/bench_next.py --only compute_loss --no-cpu-baseline --steps 20 > /dev/null 2>&1
python3 - $O <<'PY'
import sys, json, csv, glob
for l in open(sys.argv[1] + "/bench.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["config"]["workload"][:78], round(d["ms_per_step"] * 1e3, 1), "us", round(d["roofline"]["achieved"]), "GB/s",
              round(d["roofline"]["frac"], 3), d.get("cpu_baseline", {}).get("value"))
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
