#!/bin/bash
# bytes past the L2s and hit rates of the transposed tile order against the orbit order (L=32 c128)
# HISTORICAL: SD_XCD_PASC selected a tile order that existed only in commit b4b15af and was removed in b3f5209 (measured: no
# fewer bytes).  At any later commit no source reads the variable and all four passes measure the same orbit order; the records
# profiles/r03/transposed_order_counters.txt and ab_transposed_order.txt can only be reproduced from a checkout of b4b15af.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03n
mkdir -p $OUT
for k in 0 256 1024 4096; do
  if [ $k = 0 ]; then unset SD_XCD_PASC; else export SD_XCD_PASC=$k; fi
  CMD="python3 profiles/apply_once.py 32 c128 5"
  rocprofv3 --pmc FETCH_SIZE SQ_WAVES --output-format csv -d $OUT/k$k/pmc1 -- $CMD > $OUT/k$k.pmc1.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/k$k/pmc2 -- $CMD > $OUT/k$k.pmc2.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum --output-format csv -d $OUT/k$k/pmc3 -- $CMD > $OUT/k$k.pmc3.log 2>&1
  echo "=== SD_XCD_PASC=$k"; python3 profiles/summarize.py $OUT/k$k 2>&1 | grep -E "kernel void|per-dispatch" | head -40
done > $OUT/summary.txt 2>&1
cut -c1-150 $OUT/summary.txt
