#!/bin/bash
# GPU batch (round 3, call 2): full GPU suite, transfer modes, configs, f64 rows, full-basis Sz_q
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03b
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout -k 10 300 python profiles/xfer_bench.py 30 > $OUT/xfer_L30.jsonl 2> $OUT/xfer_L30.err; cat $OUT/xfer_L30.jsonl
timeout -k 10 600 python profiles/config_bench.py 2 3 4 > $OUT/configs.jsonl 2> $OUT/configs.err; cat $OUT/configs.jsonl; tail -3 $OUT/configs.err
timeout -k 10 300 python profiles/ab_env.py SD_F64_ROWS 8,4 30 2 > $OUT/f64_rows.txt 2>&1; cat $OUT/f64_rows.txt
SD_AUX_L=28 SD_AUX_FULL=1 timeout -k 10 300 python profiles/aux_bench.py > $OUT/aux_full_L28.jsonl 2> $OUT/aux_full.err; cat $OUT/aux_full_L28.jsonl
SD_SZQ_FULL_GENERIC=1 SD_AUX_L=28 SD_AUX_FULL=1 timeout -k 10 300 python profiles/aux_bench.py > $OUT/aux_full_L28_generic.jsonl 2>> $OUT/aux_full.err; head -1 $OUT/aux_full_L28_generic.jsonl
