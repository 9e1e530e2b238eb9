#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06h; rm -rf "$O"; mkdir -p "$O"
for t in 0 1 2 3; do for b in 8 2 1; do
  if [ $t = 0 ]; then LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[tile rule] /" | tee -a "$O/unet_tile_ab.txt"
  else LDIFF_GEMM_TILE=$t LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[GEMM_TILE=$t] /" | tee -a "$O/unet_tile_ab.txt"; fi
done; done
for t in 1 2; do LDIFF_GEMM_TILE=$t python3 scripts/unet_launches.py > "$O/unet_launches_b8_tile$t.txt" 2>&1; done
python3 scripts/unet_launches.py > "$O/unet_launches_b8_rule.txt" 2>&1
