#!/bin/bash
# (one TCC-heavy counter group per pass: FETCH_SIZE with TCC_HIT/MISS in one pass exceeds the hardware's counters)
# fabric-side traffic and L2 hit rate of the Float64 apply at L=30 (bash profiles/f64_counters.sh on the GPU box)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/f64ctr
mkdir -p $OUT
for dt in f64 c128; do
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$dt.p1 -- python3 profiles/apply_once.py 30 $dt > $OUT/$dt.p1.log 2>&1
  echo "$dt pass 1 done"
  timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/$dt.p2 -- python3 profiles/apply_once.py 30 $dt > $OUT/$dt.p2.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.getcwd(), "gpurun_out", "f64ctr")
for dt in ("f64", "c128"):
    tot = {}
    n = {}
    for f in glob.glob(os.path.join(out, dt + ".p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_apply_tiled" in r["Kernel_Name"]:
                k = r["Counter_Name"]
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
                n[k] = n.get(k, 0) + 1
    # 5 applies x 3 launches each: totals / 5 = per apply
    per = {k: v / 5 for k, v in tot.items()}
    N = 155117520
    fetch, write = per.get("FETCH_SIZE", 0) * 1024 * 2, per.get("WRITE_SIZE", 0) * 1024
    print(dt, {k: f"{v:.4g}" for k, v in per.items()}, "dispatches", n)
    print(dt, f"fabric-side bytes per apply {(fetch + write) / 1e9:.2f} GB = {(fetch + write) / N:.1f} B/row; L2 hit {per.get('TCC_HIT_sum', 0) / max(1.0, per.get('TCC_HIT_sum', 0) + per.get('TCC_MISS_sum', 0)):.3f}")
PY
