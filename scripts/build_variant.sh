#!/bin/bash
# diagnostic build of the whole library with extra hipcc flags into build/<name>/libldiff_hip.so (A/B runs on ONE box through LDIFF_LIB:
# boxes of the pool differ by more than most effects).  usage: scripts/build_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/$name
pids=()
for src in ldiffusion_amd/csrc/*.hip; do
  obj=build/$name/$(basename ${src%.hip}).o
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-comment "$@" -c $src -o $obj &
  pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o build/$name/libldiff_hip.so build/$name/*.o
echo built build/$name/libldiff_hip.so
