#!/bin/bash
# Same-box A/B of the UNet step (B=8, 64x64 latents, SD-v1.5 width, hipGraph replay) under an environment switch.
# usage: scripts/bench_unet_ab.sh VAR=a VAR=b ...   (each argument is one arm; "-" = no variable)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
for round in 1 2; do
  for arm in "$@"; do
    if [ "$arm" = "-" ]; then python3 scripts/unet_only.py 2>&1 | grep -E "unet step" | sed "s/^/[default] /"
    else env $arm python3 scripts/unet_only.py 2>&1 | grep -E "unet step" | sed "s/^/[$arm] /"; fi
  done
done
