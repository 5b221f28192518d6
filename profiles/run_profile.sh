#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag> [L]
# Writes rocprofv3 kernel-trace stats and PMC passes for the bench command under gpurun_out/prof_<tag>/.
set -u
TAG=${1:-r01}
L=${2:-32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --L $L --steps 50 --warmup 5 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
python3 profiles/collect_traffic.py $OUT $L c128 >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt
