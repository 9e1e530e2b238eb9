#!/bin/bash
# diagnostic build of the library with in-kernel cycle stamps in the ping-pong conv3x3 kernel (scripts/conv_stamps.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-comment -DC3P_STAMPS $C3P_EXTRA -c ldiffusion_amd/csrc/kernels_conv3x3p.hip -o build/stamps/kernels_conv3x3p.o
objs=$(ls build/obj/*.o | grep -v kernels_conv3x3p.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps/libldiff_hip.so $objs build/stamps/kernels_conv3x3p.o
echo built build/stamps/libldiff_hip.so
