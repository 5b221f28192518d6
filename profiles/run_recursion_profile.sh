#!/bin/bash
# Kernel trace of the fused recursion steps (profiles/recursion_bench.py: Chebyshev term and pair at L=32, KPM moments at L=30, Lanczos step at
# L=32, Krylov at L=28): which kernel takes how long inside a step.  bash profiles/run_recursion_profile.sh <tag>  (on the GPU box)
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_recursion_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 profiles/recursion_bench.py > $OUT/trace.log 2>&1
python3 profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
grep "^{" $OUT/trace.log >> $OUT/summary.txt
head -40 $OUT/summary.txt | cut -c1-220
