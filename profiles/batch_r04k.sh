#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04k
mkdir -p $OUT
python profiles/ab_lib.py spindynamics.jl_amd/csrc/_var/libspindyn_skipc.so 32 2 2>&1 | tee $OUT/ab_skip_dead_c128_L32.txt
python profiles/ab_lib.py spindynamics.jl_amd/csrc/_var/libspindyn_skipc.so 30 2 2>&1 | tee $OUT/ab_skip_dead_c128_L30.txt
