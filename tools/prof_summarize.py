#!/usr/bin/env python3
"""Summarise the rocprofv3 output directories written by tools/prof_run.sh: per kernel the call count and
average duration (kernel-trace stats) and the per-launch average of every collected counter."""
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]
out = {"kernels": {}}


def short(name):
    m = re.search(r"(\w+)(<[^(]*>)?\(", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        out["kernels"].setdefault(k, {})["calls"] = int(r["Calls"])
        out["kernels"][k]["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
        out["kernels"][k]["min_us"] = round(float(r["MinNs"]) / 1e3, 2)
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        meta = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            key = (k, r["Counter_Name"])
            acc.setdefault(key, []).append(float(r["Counter_Value"]))
            meta[k] = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                       "grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"])}
        for (k, c), v in acc.items():
            e = out["kernels"].setdefault(k, {})
            e.setdefault("counters", {})[c] = round(sum(v) / len(v), 3)
            e.update(meta[k])
print(json.dumps(out, indent=1))
