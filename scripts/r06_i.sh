#!/bin/bash
# where do the ~26 fillBufferAligned + ~19 copyBuffer runtime kernels per UNet pass come from: hipGraph replay or the launch sequence itself?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06k; rm -rf "$O"; mkdir -p "$O"
for m in graph eager; do
  if [ $m = eager ]; then export LDIFF_UNET_EAGER=1; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$m" -- python3 scripts/unet_only.py > "$O/$m.log" 2>&1
  f=$(find "$O/$m" -name "*kernel_stats.csv" | head -1)
  echo "== $m: $(grep 'unet step' $O/$m.log)"; grep "rocclr\|set_scalar\|zero_bytes" "$f" | cut -c1-120
  rm -rf "$O/$m"
done
