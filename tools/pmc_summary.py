#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter_collection.csv files per kernel and counter.
usage: pmc_summary.py DIR [DIR ...] [--kernel SUBSTR] [--json OUT]"""
import collections
import csv
import glob
import json
import os
import sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
kern = None
out = None
args = sys.argv[1:]
for i, a in enumerate(args):
    if a == "--kernel":
        kern = args[i + 1]
    if a == "--json":
        out = args[i + 1]
dirs = [d for d in dirs if d not in (kern, out)]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dw::" not in k or (kern and kern not in k):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, cs in res.items():
    print(k[:70])
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {v:16.1f}")
if out:
    json.dump(res, open(out, "w"), indent=1)
