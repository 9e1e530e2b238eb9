#!/bin/bash
# runtime knobs (process environment, no code): does any of them move the UNet pass?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06l; rm -rf "$O"; mkdir -p "$O"
run() { echo "[$1] $(env $1 LDIFF_UNET_B=$2 python3 scripts/unet_only.py 2>&1 | grep 'unet step')" | tee -a "$O/knobs.txt"; }
for b in 1 8; do
  run "X=0" $b
  run "HIP_FORCE_DEV_KERNARG=1" $b
  run "HIP_FORCE_DEV_KERNARG=0" $b
  run "GPU_MAX_HW_QUEUES=2" $b
  run "GPU_MAX_HW_QUEUES=8" $b
  run "HSA_ENABLE_SDMA=0" $b
  run "AMD_DIRECT_DISPATCH=0" $b
  run "HIP_LAUNCH_BLOCKING=0" $b
  run "ROC_ACTIVE_WAIT_TIMEOUT=100" $b
  run "X=1" $b
done
