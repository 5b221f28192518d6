#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03e
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_recursions.py tests/test_gpu_edge_cases.py -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 300 python profiles/groundstate_bench.py 28 100 > $OUT/groundstate.jsonl 2> $OUT/groundstate.err; cat $OUT/groundstate.jsonl; tail -2 $OUT/groundstate.err
bash profiles/run_profile.sh r03 32 > $OUT/run_profile.log 2>&1; tail -2 $OUT/run_profile.log
