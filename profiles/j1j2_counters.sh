#!/bin/bash
# fabric-side traffic, L2 hit rate and kernel times of the J1-J2 apply (general-bond plan) at L=28: bash profiles/j1j2_counters.sh on the GPU box
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/j1j2ctr
mkdir -p $OUT
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/apply_once.py 28 c128 5 j1j2 > $OUT/trace.log 2>&1
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 profiles/apply_once.py 28 c128 5 j1j2 > $OUT/p1.log 2>&1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -- python3 profiles/apply_once.py 28 c128 5 j1j2 > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.getcwd(), "gpurun_out", "j1j2ctr")
tot, n = {}, {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_apply_tiled" in r["Kernel_Name"]:
            k = r["Counter_Name"]
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
per = {k: v / 5 for k, v in tot.items()}
N = 40116600
fetch, write = per.get("FETCH_SIZE", 0) * 1024 * 2, per.get("WRITE_SIZE", 0) * 1024
print("J1-J2 L=28 c128", {k: f"{v:.4g}" for k, v in per.items()}, "dispatches", n)
print(f"fabric-side bytes per apply {(fetch + write) / 1e9:.2f} GB = {(fetch + write) / N:.1f} B/row; L2 hit {per.get('TCC_HIT_sum', 0) / max(1.0, per.get('TCC_HIT_sum', 0) + per.get('TCC_MISS_sum', 0)):.3f}")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_apply_tiled" in r["Name"]:
            print("kernel", r["Name"][:90], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
PY
