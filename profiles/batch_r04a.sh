#!/bin/bash
# round 4, packed partner table: GPU suite, timings (c128 / f64, L=30/32), counter passes
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
python profiles/variants_bench.py > $OUT/variants.jsonl 2>&1; cat $OUT/variants.jsonl
SD_F64_ROWS=4 python profiles/variants_bench.py 2>&1 | grep f64 | sed 's/^/F64_ROWS=4 /' | tee -a $OUT/variants.jsonl
python bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench_c128.json 2>$OUT/bench.err; cut -c1-400 $OUT/bench_c128.json
python bench.py --steps 20 --warmup 5 --no-cpu --dtype f64 > $OUT/bench_f64.json 2>>$OUT/bench.err; cut -c1-400 $OUT/bench_f64.json
python profiles/recursion_bench.py > $OUT/recursion.jsonl 2>&1; cat $OUT/recursion.jsonl
