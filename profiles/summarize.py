#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel-trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:70]
        print(f"  {name:70s} calls={row.get('Calls')} total_ns={row.get('TotalDurationNs')} avg_ns={row.get('AverageNs')} pct={row.get('Percentage')}")

for f in find("*kernel_trace.csv"):
    d = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[n] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
                   row.get("Workgroup_Size"), row.get("Grid_Size"))
    print("== kernel trace:", os.path.relpath(f, out))
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"  {n[:70]:70s} n={len(v)} avg={sum(v)/len(v)/1e3:.1f}us med={v2[len(v2)//2]/1e3:.1f}us min={v2[0]/1e3:.1f}us vgpr/agpr/sgpr/lds/wg/grid={meta[n]}")

for f in find("*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters:", os.path.relpath(f, out))
    for k, cs in acc.items():
        if "apply" not in k:
            continue
        print("  kernel", k[:90])
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} per-dispatch avg={sum(v)/len(v):.6g}  n={len(v)}")
