#!/bin/bash
# memory-system counters for the apply kernel: bash profiles/pmc_mem.sh <tag> <L>  (env knobs pass through)
set -u
TAG=${1:-mem}; L=${2:-32}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python3 bench.py --L $L --steps 4 --warmup 1 --no-cpu"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/pA -- $CMD > $OUT/pA.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_sum TCC_CYCLE_sum --output-format csv -d $OUT/pB -- $CMD > $OUT/pB.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_STREAMING_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pC -- $CMD > $OUT/pC.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/pD -- $CMD > $OUT/pD.log 2>&1
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
