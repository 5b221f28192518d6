#!/bin/bash
# what the periodic chain's wrap bond costs: counters of open vs periodic at L=30 (c128)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03h
mkdir -p $OUT
for bc in open periodic; do
  CMD="python3 profiles/apply_once.py 30 c128 6 $bc"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$bc/trace -- $CMD > $OUT/$bc.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/$bc/pmc1 -- $CMD > $OUT/$bc.pmc1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/$bc/pmc2 -- $CMD > $OUT/$bc.pmc2.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --output-format csv -d $OUT/$bc/pmc3 -- $CMD > $OUT/$bc.pmc3.log 2>&1
  echo "=== $bc"; python3 profiles/summarize.py $OUT/$bc 2>&1 | grep -E "kernel void|k_apply_tiled|per-dispatch" | grep -v "^  void at" | head -60
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | cut -c1-160
