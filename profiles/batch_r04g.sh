#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04g
mkdir -p $OUT
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_baseline_lengths.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -6 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
python profiles/shard_kernel_bench.py 32 8 0 2 3 > $OUT/shard_P8.jsonl 2>&1; cat $OUT/shard_P8.jsonl
