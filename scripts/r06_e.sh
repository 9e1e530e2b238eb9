#!/bin/bash
# UNet pass alone under rocprofv3: sum of kernel durations per pass against the wall time per pass (what is left is launch boundaries), B = 1 and B = 8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06g; rm -rf "$O"; mkdir -p "$O"
for b in 1 8; do
  export LDIFF_UNET_B=$b
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/b$b" -- python3 scripts/unet_only.py > "$O/unet_b$b.log" 2>&1
  f=$(find "$O/b$b" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/unet_b${b}_kernel_stats.csv"
  rm -rf "$O/b$b"
  grep "unet step" "$O/unet_b$b.log"
  python3 - "$O/unet_b${b}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows); n = sum(int(r["Calls"]) for r in rows)
print(f"  all kernels of the process: {n} launches, {tot / 1e6:.1f} ms; per pass (23 passes in the process, set-up kernels included): {n / 23:.0f} launches, {tot / 23e6:.2f} ms of kernel time")
for r in rows[:12]:
    print(f"    {r['Name'][:80]:80s} n={r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:7.1f} us  {float(r['Percentage']):5.1f} %")
PY
done
