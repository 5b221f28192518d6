#!/bin/bash
# final records of round 3: full GPU suite, rocprof summary + traffic stamp on the final kernel sources, bench line, f64 variants
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03j
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
bash profiles/run_profile.sh r03 32 > $OUT/run_profile.log 2>&1; tail -2 $OUT/run_profile.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json
timeout -k 10 300 python profiles/variants_bench.py > $OUT/variants.jsonl 2> $OUT/variants.err; cat $OUT/variants.jsonl
timeout -k 10 300 python profiles/groundstate_bench.py 28 100 > $OUT/groundstate.jsonl 2> $OUT/groundstate.err; cat $OUT/groundstate.jsonl
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
