#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04i
mkdir -p $OUT
python -m pytest tests/test_gpu_recursions.py tests/test_observables_states.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -12 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
python profiles/groundstate_bench.py 28 100 > $OUT/groundstate_L28.jsonl 2>&1; cat $OUT/groundstate_L28.jsonl
SD_GS_FUSED=0 python profiles/groundstate_bench.py 28 100 2>&1 | grep "blocks of 8" | sed 's/^/UNFUSED /' | tee -a $OUT/groundstate_L28.jsonl
python profiles/groundstate_bench.py 20 100 > $OUT/groundstate_L20.jsonl 2>&1; cat $OUT/groundstate_L20.jsonl
SD_GS_FUSED=0 python profiles/groundstate_bench.py 20 100 2>&1 | grep "blocks of 8" | sed 's/^/UNFUSED /' | tee -a $OUT/groundstate_L20.jsonl
python examples/kpm_sqw.py 2>&1 | tee $OUT/example.txt
