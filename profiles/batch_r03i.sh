#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03i
mkdir -p $OUT
PREV=spindynamics.jl_amd/libspindyn_prev.so
timeout -k 10 600 python -m pytest tests/test_gpu_apply.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded.py -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout -k 10 400 python profiles/ab_lib.py $PREV 32 2 > $OUT/ab_L32.txt 2>&1; cat $OUT/ab_L32.txt
timeout -k 10 400 python profiles/ab_lib.py $PREV 30 2 > $OUT/ab_L30.txt 2>&1; cat $OUT/ab_L30.txt
timeout -k 10 400 python profiles/ab_lib.py $PREV 28 1 > $OUT/ab_L28.txt 2>&1; cat $OUT/ab_L28.txt
