#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04f
mkdir -p $OUT
python profiles/recursion_bench.py lanczos > $OUT/lanczos.jsonl 2>&1; cat $OUT/lanczos.jsonl
python profiles/shard_kernel_bench.py 32 8 0 1 2 3 4 5 6 7 > $OUT/shard_P8.jsonl 2>&1; cat $OUT/shard_P8.jsonl
python profiles/shard_kernel_bench.py 32 4 0 1 2 3 > $OUT/shard_P4.jsonl 2>&1; cat $OUT/shard_P4.jsonl
python profiles/shard_kernel_bench.py 32 2 0 1 > $OUT/shard_P2.jsonl 2>&1; cat $OUT/shard_P2.jsonl
