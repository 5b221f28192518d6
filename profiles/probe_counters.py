#!/usr/bin/env python3
"""Per-kernel averages of the two-pass probe's rocprofv3 passes (kernel-trace + PMC): ms, 2*FETCH_SIZE + WRITE_SIZE (gfx950
half-count correction of the guide), L2 hit rate.  Usage: python profiles/probe_counters.py gpurun_out/probe"""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    if "k_p2" in n:
        return "P2 " + n[:40]
    if "k_apply_tiled" in n:
        return ("P1/full DIAG " if n.rstrip(")").endswith("true>") or "Lb1ELb1EE" in n else "apply ") + n[:60]
    return None


for f in sorted(glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)):
    d = defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("== durations:", os.path.relpath(f, out))
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if short(n):
            print(f"  {n[:100]:100s} n={len(v)} avg={sum(v)/len(v)/1e6:.3f} ms")
for f in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("== counters:", os.path.relpath(f, out))
    for k, cs in acc.items():
        if not short(k):
            continue
        print("  kernel", k[:100])
        for c, v in sorted(cs.items()):
            print(f"    {c:24s} per-dispatch avg={sum(v)/len(v):.6g}  n={len(v)}")
