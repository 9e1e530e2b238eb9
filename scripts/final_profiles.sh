#!/bin/bash
# Round-end evidence: rocprofv3 kernel stats of the bench command + PMC traffic passes.  Run on the GPU box via gpurun.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final; rm -rf "$O"; mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-unet-step > "$O/bench_under_rocprofv3.json" 2> "$O/stats.log"
f=$(find "$O/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/rocprofv3_kernel_stats.csv"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unet-step --no-prof > "$O/pmcf.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unet-step --no-prof > "$O/pmcw.log" 2>&1
H=$(python3 -c "import bench; print(bench.kernel_source_hash())")
python3 scripts/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_traffic_$H.json"   # bench.py only quotes the file of the current kernel sources
rm -rf "$O/stats" "$O/pmc_fetch" "$O/pmc_write"
# the UNet pass under the same counters (bench.py quotes it as unet_step*.pmc, by the same source hash): gpurun_out/pmc_unet_<hash>.json
rm -f "gpurun_out/pmc_unet_$H.json"
for b in 8 2 1; do bash scripts/pmc_unet_pass.sh $b > /dev/null 2>&1; done
cp "gpurun_out/pmc_unet_$H.json" "$O/" 2>/dev/null
tail -c 400 "$O/bench_under_rocprofv3.json"; head -5 "$O/rocprofv3_kernel_stats.csv" | cut -c1-200
