#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04n
mkdir -p $OUT
V=$PWD/spindynamics.jl_amd/csrc/_var
for lib in default ${1:-gen5}; do
  if [ $lib = default ]; then unset SD_LIB_PATH; else export SD_LIB_PATH=$V/libspindyn_$lib.so; fi
  echo "lib=$lib" | tee -a $OUT/general_bonds.jsonl
  python profiles/general_bonds_bench.py 28 2>/dev/null | tee -a $OUT/general_bonds.jsonl
  python profiles/general_bonds_bench.py 24 2>/dev/null | grep "all pairs" | tee -a $OUT/general_bonds.jsonl
done
