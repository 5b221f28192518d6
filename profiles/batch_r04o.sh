#!/bin/bash
# skip the loads of dead row groups for ComplexF64 in the one-wave (short-tile) workgroups only: A/B on dilute and half-filled sectors
set -u
V=$PWD/spindynamics.jl_amd/csrc/_var/libspindyn_skipshort.so
for cfg in "36 9" "32 8" "34 12" "32 16" "30 15"; do
  set -- $cfg
  for lib in base skip base skip; do
    if [ $lib = skip ]; then export SD_LIB_PATH=$V; else unset SD_LIB_PATH; fi
    timeout -k 10 200 python profiles/dilute_sector_bench.py $1 $2 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['L'], d['nup'], d['path'], d['ms'], d['Grows_per_s'])"
  done
done
