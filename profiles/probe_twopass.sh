#!/bin/bash
# Usage (GPU box, repo root): bash profiles/probe_twopass.sh   -> gpurun_out/probe/*.jsonl, counters under gpurun_out/probe/pmc*
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/probe
mkdir -p $OUT
python3 profiles/probe_twopass.py 30 10 128,256,512 20 > $OUT/L30_m10.jsonl 2> $OUT/L30_m10.err || { tail -5 $OUT/L30_m10.err; exit 1; }
cat $OUT/L30_m10.jsonl
python3 profiles/probe_twopass.py 32 10 128,256,512,1024 20 > $OUT/L32_m10.jsonl 2> $OUT/L32_m10.err || { tail -5 $OUT/L32_m10.err; exit 1; }
cat $OUT/L32_m10.jsonl
python3 profiles/probe_twopass.py 32 8 256 20 > $OUT/L32_m8.jsonl 2> $OUT/L32_m8.err || { tail -5 $OUT/L32_m8.err; exit 1; }
cat $OUT/L32_m8.jsonl
python3 profiles/probe_twopass.py 32 11 128,256 20 > $OUT/L32_m11.jsonl 2> $OUT/L32_m11.err || { tail -5 $OUT/L32_m11.err; exit 1; }
cat $OUT/L32_m11.jsonl
CMD="python3 profiles/probe_twopass.py 32 10 256 3 counters"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
python3 profiles/probe_counters.py $OUT > $OUT/counters.txt 2>&1
cat $OUT/counters.txt
