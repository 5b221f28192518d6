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

# registers / LDS / scratch come from the CODE OBJECT (profiles/codeobj_meta.py), not from the trace columns: rocprofv3's
# VGPR_Count is not the kernel's allocation and its LDS_Block_Size misses dynamic LDS (evidence hygiene 9)
try:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import codeobj_meta
    CO = codeobj_meta.meta()
except Exception as e:      # tools missing: say so, never print the misleading columns instead
    CO = {}
    print("(code-object metadata unavailable: %r)" % (e,))


def _norm(n):
    return n.replace("kernel ", "").replace("void ", "").strip()


def co_of(trace_name):
    key = _norm(trace_name)
    for n, d in CO.items():
        if _norm(n) == key:
            return d
    for n, d in CO.items():          # a truncated trace name
        if len(key) > 24 and (_norm(n).startswith(key) or key.startswith(_norm(n))):
            return d
    return None


for f in find("*kernel_trace.csv"):
    d = defaultdict(list)
    meta = {}
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        d[n].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        meta[n] = (row.get("Workgroup_Size"), row.get("Grid_Size"))
    print("== kernel trace:", os.path.relpath(f, out), "(vgpr/sgpr/static-LDS/scratch/waves-per-SIMD from the code object;")
    print("   k_apply_tiled adds DYNAMIC LDS per launch: (max_len+1)*sizeof(V) + 16*17*4 + 256 B = 16.1 KiB for 924-row c128 tiles)")
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        c = co_of(n)
        regs = ("vgpr=%d agpr=%d sgpr=%d lds_static=%d scratch=%d waves/SIMD<=%d" % (
            c.get("vgpr", -1), c.get("agpr", 0), c.get("sgpr", -1), c.get("lds_static", 0), c.get("scratch", 0),
            codeobj_meta.waves_per_simd(c.get("vgpr", 512), c.get("agpr", 0)))) if c else "code object: not found"
        print(f"  {n[:70]:70s} n={len(v)} avg={sum(v)/len(v)/1e3:.1f}us med={v2[len(v2)//2]/1e3:.1f}us min={v2[0]/1e3:.1f}us {regs} wg/grid={meta[n]}")

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
