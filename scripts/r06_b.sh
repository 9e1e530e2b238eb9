#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; rm -rf "$O"; mkdir -p "$O"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_groupnorm or test_conv or split" > "$O/pytest_kernels.log" 2>&1; tail -3 "$O/pytest_kernels.log"
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "tiny or overflow or graph or precision" > "$O/pytest_models.log" 2>&1; tail -3 "$O/pytest_models.log"
for fine in 0 1; do for b in 1 2 8; do LDIFF_SPLITK_FINE=$fine LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed "s/^/[FINE=$fine] /" | tee -a "$O/unet_ab.txt"; done; done
LDIFF_UNET_B=1 python3 scripts/unet_launches.py > "$O/unet_launches_b1.txt" 2>&1
python3 scripts/unet_launches.py > "$O/unet_launches_b8.txt" 2>&1
head -5 "$O/unet_launches_b1.txt"
