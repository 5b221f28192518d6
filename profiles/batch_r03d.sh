#!/bin/bash
# GPU batch (round 3, call 4): the records of the round -- GPU suite, bench line with cpu_baseline, rocprof summary + traffic,
# recursion / variant timings
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03d
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
bash profiles/run_profile.sh r03 32 > $OUT/run_profile.log 2>&1; tail -5 $OUT/run_profile.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
timeout -k 10 600 python profiles/recursion_bench.py > $OUT/recursion.jsonl 2> $OUT/recursion.err; cat $OUT/recursion.jsonl
timeout -k 10 300 python profiles/variants_bench.py > $OUT/variants.jsonl 2> $OUT/variants.err; cat $OUT/variants.jsonl
