#!/bin/bash
# round 6: GroupNorm transform spread over taps 1..8 (conv3x3w) against the build without it (build/nospread), same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06j; rm -rf "$O"; mkdir -p "$O"
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or groupnorm or splitk or wide" 2>&1 | tail -5 | tee "$O/conv_tests.txt"
python3 scripts/conv_stamps_w.py > "$O/c3w_stamps_spread.txt" 2>&1; cat "$O/c3w_stamps_spread.txt"
for r in 1 2; do for b in 8 1; do
  LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[spread] /" | tee -a "$O/unet_spread_ab.txt"
  LDIFF_LIB=build/nospread/libldiff_hip.so LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[before] /" | tee -a "$O/unet_spread_ab.txt"
done; done
