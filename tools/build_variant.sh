#!/bin/bash
# Developer tool: build libobia_hip_<name>.so with extra -D flags for slic_sweep.hip only (kernel experiments).
#   tools/build_variant.sh w5 -DASSIGN_WAVES=5
# Run a bench against it with  OBIA_HIP_LIB=obia_amd/csrc/libobia_hip_w5.so python bench.py ...
set -e
name=$1; shift
cd "$(dirname "$0")/../obia_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 -Wall -Wno-unused-result "$@" -c slic_sweep.hip -o slic_sweep_$name.o 2>&1 | grep -E "error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libobia_hip_$name.so context.o api.o slic.o slic_sweep_$name.o cc.o zonal.o tiling.o quickshift.o polygons.o consumers.o texture.o
echo built libobia_hip_$name.so
