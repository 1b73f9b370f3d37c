#!/bin/bash
# Developer tool: build libobia_hip_<name>.so with extra -D flags for ONE source file (kernel experiments).
#   tools/build_variant.sh w5 -DASSIGN_WAVES=5                 (slic_sweep.hip, the default)
#   SRC=zonal tools/build_variant.sh z4 -DZW=4                 (another source)
# Run a bench against it with  OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_w5.so python bench.py ...
set -e
name=$1; shift
src=${SRC:-slic_sweep}
cd "$(dirname "$0")/../obia_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wall -Wno-unused-result "$@" -c $src.hip -o ${src}_$name.o 2>&1 | grep -E "error" || true
objs=""
for o in context api slic slic_sweep cc zonal tiling quickshift polygons consumers texture; do
    if [ "$o" = "$src" ]; then objs="$objs ${src}_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libobia_hip_$name.so $objs
echo built libobia_hip_$name.so
# the 64-bit-shift erratum guard of the regular build (tools/check_shift64.py) covers variant libraries too
python3 ../../tools/check_shift64.py libobia_hip_$name.so | grep -E "BAD|EMPTY" || true
