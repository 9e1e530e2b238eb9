#!/bin/bash
# round 6, last sources (no test run: scripts/r06_final.sh ran the suite on the same kernels): profiles + bench line + per-launch lists
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06y; rm -rf "$O"; mkdir -p "$O"
bash scripts/final_profiles.sh > "$O/final_profiles.log" 2>&1
python3 bench.py > "$O/bench_run1.json" 2> "$O/bench_run1.err"; tail -c 300 "$O/bench_run1.json"
python3 bench.py > "$O/bench_run2.json" 2> "$O/bench_run2.err"
python3 scripts/unet_launches.py > "$O/unet_launches.txt" 2>&1
LDIFF_UNET_B=1 python3 scripts/unet_launches.py > "$O/unet_launches_b1.txt" 2>&1
ls gpurun_out/final | grep -v pmc_traffic_ ; ls gpurun_out/final | grep 29dec
