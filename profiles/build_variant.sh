#!/bin/bash
# Build an A/B variant of libspindyn.so with extra -D flags for kernels_apply.hip:  bash profiles/build_variant.sh <name> -DFOO=1 ...
# -> spindynamics.jl_amd/csrc/_var/libspindyn_<name>.so (git-ignored; travels to the GPU box); compare with profiles/ab_lib.py
set -e
NAME=$1; shift
cd "$(dirname "$0")/../spindynamics.jl_amd/csrc"
make -s -j8
mkdir -p _var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function "$@" -c kernels_apply.hip -o _var/kernels_apply_$NAME.o
OBJS=$(ls _obj/*.o | grep -v kernels_apply)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS _var/kernels_apply_$NAME.o -o _var/libspindyn_$NAME.so
echo built _var/libspindyn_$NAME.so
