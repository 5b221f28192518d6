#!/bin/bash
# general-bond plan (k_apply_tiled GEN): bit-exact tests, then A/B against the per-row form
set -u
OUT=$PWD/gpurun_out/r04m
mkdir -p $OUT
python -m pytest tests/test_gpu_apply.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_sharded.py tests/test_gpu_fullsize_random.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -15 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
for plan in 1 0; do
  echo "SD_GEN_PLAN=$plan" | tee -a $OUT/general_bonds.jsonl
  SD_GEN_PLAN=$plan python profiles/general_bonds_bench.py 28 2>/dev/null | tee -a $OUT/general_bonds.jsonl
  SD_GEN_PLAN=$plan python profiles/general_bonds_bench.py 24 2>/dev/null | tee -a $OUT/general_bonds.jsonl
done
