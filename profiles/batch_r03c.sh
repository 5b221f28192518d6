#!/bin/bash
# GPU batch (round 3, call 3): counted-wait far-bond pipeline A/B against the previous kernel build, wrap-aware tile order,
# full-basis Sz_q, config timings, full GPU suite
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03c
mkdir -p $OUT
PREV=spindynamics.jl_amd/libspindyn_prev.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
timeout -k 10 400 python profiles/ab_lib.py $PREV 30 2 > $OUT/ab_L30.txt 2>&1; cat $OUT/ab_L30.txt
timeout -k 10 400 python profiles/ab_lib.py $PREV 32 1 > $OUT/ab_L32.txt 2>&1; cat $OUT/ab_L32.txt
timeout -k 10 400 python profiles/ab_lib.py $PREV 28 1 > $OUT/ab_L28.txt 2>&1; cat $OUT/ab_L28.txt
for lib in "" $PREV; do SD_LIB_PATH=$lib timeout -k 10 300 python profiles/couplings_bench.py 30 >> $OUT/couplings.jsonl 2>> $OUT/couplings.err; echo "--- lib=$lib" >> $OUT/couplings.jsonl; done; cat $OUT/couplings.jsonl
for w in 1 0 1 0; do SD_XCD_WRAP=$w timeout -k 10 200 python profiles/periodic_ab.py 28 | sed "s/^/wrap=$w /" >> $OUT/periodic.txt 2>> $OUT/periodic.err; done
for w in 1 0; do SD_XCD_WRAP=$w timeout -k 10 200 python profiles/periodic_ab.py 30 | sed "s/^/wrap=$w /" >> $OUT/periodic.txt 2>> $OUT/periodic.err; done; cat $OUT/periodic.txt
SD_AUX_L=28 SD_AUX_FULL=1 timeout -k 10 300 python profiles/aux_bench.py > $OUT/aux_full_L28.jsonl 2> $OUT/aux_full.err; head -1 $OUT/aux_full_L28.jsonl
timeout -k 10 600 python profiles/config_bench.py 2 4 > $OUT/configs.jsonl 2> $OUT/configs.err; cat $OUT/configs.jsonl
timeout -k 10 300 python bench.py --steps 30 --no-cpu > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
