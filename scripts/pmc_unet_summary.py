"""Per-kernel PMC table of a UNet pass: scripts/pmc_unet_pass.sh runs scripts/unet_only.py (eager launches: one dispatch per kernel) under three
rocprofv3 --pmc passes; this sums each counter over all rows of a dispatch (one row per counter instance), groups the dispatches by kernel and prints,
per UNet pass: launches, kernel time, MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), HBM bytes =
2 x FETCH_SIZE + WRITE_SIZE (KB; the x 2: MI355X_MICROARCH.md and profiles/r06_fetch_size_calibration.txt) and the rate they give.
usage: python scripts/pmc_unet_summary.py <dir with busy/ fetch/ write/> <passes in the process> [<batch> <out.json>: the whole-pass row is merged into out.json under the batch]"""
import csv, glob, os, re, sys
from collections import defaultdict
root, npass = sys.argv[1], int(sys.argv[2])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:44]


def collect(sub):
    disp = defaultdict(lambda: defaultdict(float))
    meta = {}
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (f, r["Dispatch_Id"])
                disp[key][r["Counter_Name"]] += float(r["Counter_Value"])
                if key not in meta:
                    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if r.get("Start_Timestamp") and r.get("End_Timestamp") else 0.0
                    meta[key] = (short(r["Kernel_Name"]), dur)
    out = defaultdict(lambda: defaultdict(float))
    for key, c in disp.items():
        name, dur = meta[key]
        out[name]["n"] += 1
        out[name]["us"] += dur
        for k, v in c.items():
            out[name][k] += v
    return out


busy, fetch, write = collect("busy"), collect("fetch"), collect("write")
names = sorted(busy, key=lambda n: -busy[n]["us"])
tot = defaultdict(float)
print(f"{'kernel':44s} {'n/pass':>7s} {'us/pass':>9s} {'mfma busy':>9s} {'MB/pass':>9s} {'GB/s':>7s}")
for n in names:
    b = busy[n]
    if b["n"] < npass * 0.5:   # set-up kernels (weight packing etc.), not part of a pass
        continue
    simd_cycles = 1024.0 * b["GRBM_GUI_ACTIVE"] / 8.0
    frac = b["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles if simd_cycles else 0.0
    kb = 2.0 * fetch[n]["FETCH_SIZE"] + write[n]["WRITE_SIZE"]
    us_fw = 0.5 * (fetch[n]["us"] + write[n]["us"]) or b["us"]
    print(f"{n:44s} {b['n'] / npass:7.1f} {b['us'] / npass:9.1f} {frac:9.3f} {kb * 1024 / 1e6 / npass:9.1f} {kb * 1024 / 1e3 / us_fw if us_fw else 0:7.0f}")
    tot["us"] += b["us"]; tot["busy"] += b["SQ_VALU_MFMA_BUSY_CYCLES"]; tot["cyc"] += simd_cycles; tot["kb"] += kb; tot["us_fw"] += us_fw; tot["n"] += b["n"]
print(f"{'whole pass (sum over its kernels)':44s} {tot['n'] / npass:7.1f} {tot['us'] / npass:9.1f} {tot['busy'] / tot['cyc']:9.3f} {tot['kb'] * 1024 / 1e6 / npass:9.1f} {tot['kb'] * 1024 / 1e3 / tot['us_fw']:7.0f}")
print("(mfma busy: share of the dispatch's SIMD-cycles in which the matrix pipe was busy, under the profiler's clocks; GB/s: PMC bytes over the kernels' own durations in the fetch / write passes)")
if len(sys.argv) > 4:   # the whole-pass figures for bench.py (profiles/pmc_unet_<kernel source hash>.json)
    import json
    batch, out = sys.argv[3], sys.argv[4]
    try:
        doc = json.load(open(out))
    except (OSError, ValueError):
        doc = {"_note": "scripts/pmc_unet_pass.sh: the UNet pass (eager launches) under three separate rocprofv3 --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; "
                        "FETCH_SIZE; WRITE_SIZE), summed over the kernels of a pass.  mfma_busy = busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); memory_side_bytes = "
                        "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per pass: what crosses the memory side of the eight L2s (Infinity-Cache hits included), over the kernels' own "
                        "durations in those passes = memory_side_GBps", "batches": {}}
    doc["batches"][batch] = {"mfma_busy": tot["busy"] / tot["cyc"], "memory_side_bytes": tot["kb"] * 1024 / npass,
                             "memory_side_GBps": tot["kb"] * 1024 / 1e3 / tot["us_fw"], "kernel_us_per_pass": tot["us"] / npass, "launches_per_pass": tot["n"] / npass}
    json.dump(doc, open(out, "w"), indent=1)
