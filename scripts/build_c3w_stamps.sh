#!/bin/bash
# diagnostic build of the library with in-kernel cycle stamps in the 8x16-tile conv3x3 kernel (scripts/conv_stamps_w.py); never the product library
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps_c3w
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wno-comment -DC3W_STAMPS -c ldiffusion_amd/csrc/kernels_conv3x3.hip -o build/stamps_c3w/kernels_conv3x3.o
objs=$(ls build/obj/*.o | grep -v "/kernels_conv3x3.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamps_c3w/libldiff_hip.so $objs build/stamps_c3w/kernels_conv3x3.o
echo built build/stamps_c3w/libldiff_hip.so
