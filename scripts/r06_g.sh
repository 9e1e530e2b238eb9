#!/bin/bash
# round 6: the 160-column conv kernel's three-slot weight ring against the two-slot build (build/ring2), same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06i; rm -rf "$O"; mkdir -p "$O"
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv or groupnorm or splitk or wide" 2>&1 | tail -5 | tee "$O/conv_tests.txt"
for r in 1 2; do for b in 8 1; do
  LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[ring3] /" | tee -a "$O/unet_ring_ab.txt"
  LDIFF_LIB=build/ring2/libldiff_hip.so LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[ring2] /" | tee -a "$O/unet_ring_ab.txt"
done; done
python3 scripts/unet_launches.py 2>&1 | grep "160" | tee "$O/launches_160_ring3.txt"
