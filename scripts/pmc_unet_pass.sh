#!/bin/bash
# PMC table of the UNet pass (north_star: "rocprof-reported HBM GB/s and MFMA utilisation against gfx950 peak" for the UNet step): three separate --pmc passes
# (kernel-trace only) over scripts/unet_only.py with eager launches, so that every kernel of a pass is a dispatch of its own.  usage: bash scripts/pmc_unet_pass.sh [batch]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp LDIFF_UNET_EAGER=1 LDIFF_UNET_B=${1:-8}
O=gpurun_out/pmc_unet_b$LDIFF_UNET_B; rm -rf "$O"; mkdir -p "$O"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/busy" -- python3 scripts/unet_only.py > "$O/busy.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -- python3 scripts/unet_only.py > "$O/fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -- python3 scripts/unet_only.py > "$O/write.log" 2>&1
{ echo "UNet pass, B = $LDIFF_UNET_B, eager launches under rocprofv3 --pmc (23 passes in the process):"; grep -h "unet step" "$O"/busy.log "$O"/fetch.log "$O"/write.log; python3 scripts/pmc_unet_summary.py "$O" 23 $LDIFF_UNET_B "gpurun_out/pmc_unet_$(python3 -c "import bench; print(bench.kernel_source_hash())").json"; } > "$O/summary.txt" 2>&1
rm -rf "$O/busy" "$O/fetch" "$O/write"
cat "$O/summary.txt"
