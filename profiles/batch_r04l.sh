#!/bin/bash
# Float64 row pairs (k_apply_tiled PAIRS): bit-exact tests, then A/B on one box
set -u
OUT=$PWD/gpurun_out/r04l
mkdir -p $OUT
python -m pytest tests/test_gpu_apply.py tests/test_gpu_sharded.py tests/test_gpu_recursions.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
V=$PWD/spindynamics.jl_amd/csrc/_var/libspindyn_lb6.so
for L in 30 32; do
  for pairs in 1 0 6 1 0 6; do
    if [ $pairs = 6 ]; then export SD_LIB_PATH=$V; else unset SD_LIB_PATH; fi
    SD_F64_PAIRS=$pairs python bench.py --L $L --steps 20 --warmup 5 --no-cpu --dtype f64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L=$L pairs=$pairs ms', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4))" | tee -a $OUT/ab.txt
  done
done
