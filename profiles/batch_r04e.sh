#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04e
mkdir -p $OUT
python -m pytest tests/test_gpu_apply.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py tests/test_gpu_edge_cases.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -8 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
for L in 28 30; do
  python profiles/periodic_ab.py $L 2>&1 | tee -a $OUT/periodic.txt
  SD_NO_WRAP_IMAGE=1 python profiles/periodic_ab.py $L 2>&1 | sed 's/^/GATHER /' | tee -a $OUT/periodic.txt
done
