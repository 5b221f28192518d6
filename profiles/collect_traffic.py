#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE PMC passes of profiles/run_profile.sh into profiles/traffic_latest.json.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads
-> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Units: KiB."""
import csv
import glob
import json
import os
import sys

out, L, dtype = sys.argv[1], int(sys.argv[2]), sys.argv[3]
# one apply = one launch of k_apply_tiled per tile length class (different template instantiations = different kernel
# names): average each instantiation over its dispatches, then add the instantiations up
vals = {"FETCH_SIZE": {}, "WRITE_SIZE": {}, "SQ_INSTS_VALU": {}}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_apply_tiled" in r["Kernel_Name"] and r["Counter_Name"] in vals:
            vals[r["Counter_Name"]].setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
fetch = sum(sum(v) / len(v) for v in vals["FETCH_SIZE"].values())
write = sum(sum(v) / len(v) for v in vals["WRITE_SIZE"].values())
valu = sum(sum(v) / len(v) for v in vals["SQ_INSTS_VALU"].values())       # wave-instructions per apply (run_profile.sh, pmc1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench     # kernel_source_hash(): bench.py reports the figure only for the sources it was measured on
res = {"L": L, "dtype": dtype, "source_hash": bench.kernel_source_hash(), "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write,
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "SQ_INSTS_VALU_per_launch": valu or None, "launches_per_apply": len(vals["FETCH_SIZE"]),
       "note": "FETCH_SIZE doubled per the gfx950 half-count of wide coalesced reads; this is traffic on the fabric side of the "
               "L2s (TCC_EA requests): Infinity-Cache hits are included, so it bounds HBM traffic from above"}
json.dump(res, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic_latest.json"), "w"), indent=1)
# the GPU box only hands back gpurun_out/: leave a copy there to be committed as profiles/traffic_latest.json
json.dump(res, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
print(res)
