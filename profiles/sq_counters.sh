#!/bin/bash
# SQ-side counters of the apply for Float64 and ComplexF64 at L=30 (bash profiles/sq_counters.sh on the GPU box)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/sqctr
mkdir -p $OUT
for dt in f64 c128; do
  timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/$dt.p1 -- python3 profiles/apply_once.py 30 $dt > $OUT/$dt.p1.log 2>&1
  echo "$dt pass 1 done"
  timeout -k 10 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/$dt.p2 -- python3 profiles/apply_once.py 30 $dt > $OUT/$dt.p2.log 2>&1
  echo "$dt pass 2 done"
done
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.getcwd(), "gpurun_out", "sqctr")
for dt in ("f64", "c128"):
    tot = {}
    for f in glob.glob(os.path.join(out, dt + ".p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_apply_tiled" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    per = {k: v / 5 for k, v in tot.items()}        # 5 applies
    N = 155117520
    print(dt, {k: f"{v:.4g}" for k, v in sorted(per.items())})
    g = per.get("GRBM_GUI_ACTIVE", 0) / 8
    if g:
        print(dt, f"VALUBusy {per.get('SQ_ACTIVE_INST_VALU', 0) * 4 / 1024 / g:.3f}  VALU wave-instr/row {per.get('SQ_INSTS_VALU', 0) * 64 / N:.0f} lane-ops"
              f"  LDS instr/row {per.get('SQ_INSTS_LDS', 0) * 64 / N:.1f}  VMEM_RD instr/row {per.get('SQ_INSTS_VMEM_RD', 0) * 64 / N:.1f}"
              f"  SALU/row {per.get('SQ_INSTS_SALU', 0) * 64 / N:.1f}  wave-cycles per row {per.get('SQ_WAVE_CYCLES', 0) * 4 / N:.1f}"
              f"  wait_any/wave_cycles {per.get('SQ_WAIT_ANY', 0) / max(per.get('SQ_WAVE_CYCLES', 1), 1):.3f}"
              f"  wait_inst_any/wave_cycles {per.get('SQ_WAIT_INST_ANY', 0) / max(per.get('SQ_WAVE_CYCLES', 1), 1):.3f}")
PY
