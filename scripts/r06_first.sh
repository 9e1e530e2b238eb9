#!/bin/bash
# round 6, first GPU call: micro-benchmarks (HBM ring depth, FETCH_SIZE calibration), the new tests, baseline UNet pass at B = 1 / 8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06a; rm -rf "$O"; mkdir -p "$O"
timeout 300 ./scripts/micro/hbm_ring > "$O/hbm_ring.txt" 2>&1
timeout 300 ./scripts/micro/fetch_calib > "$O/fetch_calib_plain.txt" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/calib_f" -- ./scripts/micro/fetch_calib > "$O/calib_f.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/calib_w" -- ./scripts/micro/fetch_calib > "$O/calib_w.log" 2>&1
for d in calib_f calib_w; do f=$(find "$O/$d" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cut -d, -f1-20 "$f" | python3 -c "
import csv,sys
for r in csv.DictReader(sys.stdin): print(r.get('Kernel_Name','')[:40], r.get('Counter_Name'), r.get('Counter_Value'))" > "$O/$d.txt"; rm -rf "$O/$d"; done
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -s -k "overflow or tiny or decode_latents" > "$O/pytest_subset.log" 2>&1; tail -3 "$O/pytest_subset.log"
for b in 1 2 8; do LDIFF_UNET_B=$b python3 scripts/unet_only.py 2>&1 | grep "unet step" | tee -a "$O/unet_base.txt"; done
LDIFF_UNET_B=1 LDIFF_GEMM_DF=0 python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed 's/^/[GEMM_DF=0] /' | tee -a "$O/unet_base.txt"
LDIFF_UNET_B=2 LDIFF_GEMM_DF=0 python3 scripts/unet_only.py 2>&1 | grep "unet step" | sed 's/^/[GEMM_DF=0] /' | tee -a "$O/unet_base.txt"
