#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04j
mkdir -p $OUT
python -m pytest tests/test_gpu_recursions.py tests/test_gpu_baseline_lengths.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
python profiles/smallL_bench.py > $OUT/smallL.jsonl 2>&1; cut -c1-220 $OUT/smallL.jsonl
python profiles/smallL_sqw_bench.py > $OUT/smallL_sqw.jsonl 2>&1; cat $OUT/smallL_sqw.jsonl
python examples/kpm_sqw.py 2>&1 | tee $OUT/example.txt
