#!/bin/bash
# PMC counters of the level-0 self-attention launch, generic against fixed-reference kernel (two counter groups per kernel: separate passes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/pmc_attn; rm -rf "$O"; mkdir -p "$O"
for k in 0 1; do
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/a$k" -- python3 scripts/pmc_attn_probe.py $k > "$O/a$k.log" 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d "$O/b$k" -- python3 scripts/pmc_attn_probe.py $k > "$O/b$k.log" 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for tag in ("a0", "b0", "a1", "b1"):
    fs = glob.glob(f"gpurun_out/pmc_attn/{tag}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "attn" not in name: continue
            acc[name[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(name[:60], r["Counter_Name"])] += 1
    for name, d in acc.items():
        print(tag, name, {c: round(v / n[(name, c)]) for c, v in d.items()})
PY
