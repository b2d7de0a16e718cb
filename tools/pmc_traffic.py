#!/usr/bin/env python3
"""Folds the counter summary of one workload (tools/pmc_summary.py --json) into
profiles/traffic_latest.json, the file bench.py reads `roofline.traffic` and the
vector-issue limit from.  The file is stamped with a hash of the sources the tile
kernel is built from; bench.py ignores it when the hash differs from the sources
of the library it runs.

usage: pmc_traffic.py SUMMARY.json B H W [--source profiles/NAME.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha  # noqa: E402

summary, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else summary
res = json.load(open(summary))
k = next(v for name, v in res.items() if "photo_tile_kernel<true>" in name)
path = os.path.join(ROOT, "profiles", "traffic_latest.json")
sha = kernel_sources_sha()
doc = {}
if os.path.exists(path):
    doc = json.load(open(path))
if doc.get("sources_sha") != sha:
    doc = {"kernel": "photo_tile_kernel<true>", "sources_sha": sha, "workloads": {}}
doc["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB), averaged per launch; "
               "bytes_corrected = 2 * FETCH + WRITE (gfx950 FETCH_SIZE counts half of a wide coalesced "
               "stream; this kernel's dword gathers are uncalibrated, so the true figure lies between raw "
               "and corrected).  The counters include Infinity-Cache hits.  SQ_INSTS_VALU: vector "
               "instructions issued per launch (wave granularity).  sources_sha: sha256 of "
               "csrc/photometric.hip + csrc/dw_common.h the counters were measured on.")
fetch, write = k["FETCH_SIZE"], k["WRITE_SIZE"]
doc["workloads"][f"{B}x{H}x{W}"] = {
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "bytes_raw": 1024.0 * (fetch + write), "bytes_corrected": 1024.0 * (2 * fetch + write),
    "SQ_INSTS_VALU": k.get("SQ_INSTS_VALU"), "source": source}
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(doc["workloads"][f"{B}x{H}x{W}"]))
