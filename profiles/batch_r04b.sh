#!/bin/bash
# round 4: occupancy A/B (7 waves, one / two stream register sets) and counters of the packed-table kernel
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04b
mkdir -p $OUT
V=spindynamics.jl_amd/csrc/_var
for L in 30 32; do
  for v in d2w7 d1w7; do
    python profiles/ab_lib.py $V/libspindyn_$v.so $L 2 2>&1 | tee -a $OUT/ab_L$L.txt
  done
done
bash profiles/run_profile.sh r04b 32 > $OUT/profile.log 2>&1
cp gpurun_out/prof_r04b/summary.txt $OUT/rocprof_summary.txt; cp gpurun_out/prof_r04b/traffic_latest.json $OUT/ 2>/dev/null
grep -E "SQ_INSTS_VALU|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|avg=" $OUT/rocprof_summary.txt | head -60
tail -3 $OUT/rocprof_summary.txt
