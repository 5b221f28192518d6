#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04h
mkdir -p $OUT
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_baseline_lengths.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -4 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
python profiles/shard_kernel_bench.py 32 8 0 1 2 3 4 5 6 7 > $OUT/shard_P8.jsonl 2>&1; cut -c1-420 $OUT/shard_P8.jsonl
python profiles/shard_kernel_bench.py 32 4 0 1 2 3 > $OUT/shard_P4.jsonl 2>&1; cut -c1-420 $OUT/shard_P4.jsonl
python profiles/shard_kernel_bench.py 32 2 0 1 > $OUT/shard_P2.jsonl 2>&1; cut -c1-420 $OUT/shard_P2.jsonl
python profiles/shard_kernel_bench.py 36 8 0 2 > $OUT/shard_L36_P8.jsonl 2>&1; cut -c1-420 $OUT/shard_L36_P8.jsonl
