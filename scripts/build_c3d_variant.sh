#!/bin/bash
# diagnostic build that recompiles ONLY the dataflow conv kernel with extra flags and links it with the product objects (build/obj):
# build/<name>/libldiff_hip.so for LDIFF_LIB A/B runs.  usage: scripts/build_c3d_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/$name
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-comment "$@" -c ldiffusion_amd/csrc/kernels_conv3x3d.hip -o build/$name/kernels_conv3x3d.o
objs=$(ls build/obj/*.o | grep -v kernels_conv3x3d.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o build/$name/libldiff_hip.so build/$name/kernels_conv3x3d.o $objs
echo built build/$name/libldiff_hip.so
