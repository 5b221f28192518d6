#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03g
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_observables_states.py -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
for c in 0 1; do
  if [ $c = 1 ]; then export SD_OBS_CHUNKED=1; else unset SD_OBS_CHUNKED; fi
  SD_AUX_L=32 timeout -k 10 300 python profiles/aux_bench.py 2>> $OUT/aux.err | head -3 | sed "s/^/chunked=$c /" >> $OUT/aux_obs.jsonl
  SD_AUX_L=28 SD_AUX_FULL=1 timeout -k 10 300 python profiles/aux_bench.py 2>> $OUT/aux.err | head -3 | sed "s/^/chunked=$c /" >> $OUT/aux_obs.jsonl
done
cat $OUT/aux_obs.jsonl
