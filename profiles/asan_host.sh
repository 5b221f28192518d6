#!/bin/bash
# Host-side sanitizer run of the CPU test suite (no GPU needed; GPU ASAN is not available on this pool).
# SAN=address (default) or SAN=undefined.
# The five host translation units are rebuilt with -Xarch_host -fsanitize=address and linked with the regular device objects.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SAN=${SAN:-address}
OUT=${1:-/tmp/sd_$SAN}
if [ "$SAN" = address ]; then RTN=asan; else RTN=ubsan_standalone; fi
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.$RTN-x86_64.so | head -1)
mkdir -p "$OUT"
make -C "$ROOT/spindynamics.jl_amd/csrc" -j8 > /dev/null
cd "$ROOT/spindynamics.jl_amd/csrc"
for f in capi basis recur comm xfer; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Xarch_host -fsanitize=$SAN \
    -Xarch_host -fno-omit-frame-pointer -x hip -c $f.cpp -o "$OUT/$f.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=$SAN -shared-libsan "$OUT"/capi.o "$OUT"/basis.o "$OUT"/recur.o \
  "$OUT"/comm.o "$OUT"/xfer.o _obj/kernels_apply.hip.o _obj/kernels_aux.hip.o _obj/kernels_blas1.hip.o -o "$OUT/libspindyn_asan.so"
cd "$ROOT"
SD_LIB_PATH="$OUT/libspindyn_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 SD_NO_TORCH_PRELOAD=1 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
