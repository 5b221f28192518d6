#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04c
mkdir -p $OUT
python -m pytest tests/test_gpu_recursions.py tests/test_gpu_baseline_lengths.py tests/test_gpu_edge_cases.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log
python profiles/smallL_bench.py > $OUT/smallL.jsonl 2>&1; cat $OUT/smallL.jsonl
SD_LANCZOS_FUSED=0 python profiles/smallL_bench.py 2>&1 | grep lanczos_tridiag | sed 's/^/UNFUSED /' | tee -a $OUT/smallL.jsonl
python profiles/smallL_sqw_bench.py > $OUT/smallL_sqw.jsonl 2>&1; cat $OUT/smallL_sqw.jsonl
python examples/kpm_sqw.py 2>&1 | tee $OUT/example_kpm_sqw.txt
