"""Summarise rocprofv3 --pmc output PER DISPATCH: for kernels whose name contains a substring, the counter values summed over all rows of a
dispatch (rocprofv3 writes one row per counter instance / dimension), averaged over the dispatches of each grid size, beside the dispatch's duration.
usage: python scripts/pmc_summary2.py <dir> <kernel-substring>"""
import csv, glob, os, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2]
disp = defaultdict(lambda: defaultdict(float))   # (file, dispatch id) -> counter -> sum
meta = {}
cols = None
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        cols = cols or rd.fieldnames
        for r in rd:
            if sub not in r["Kernel_Name"]:
                continue
            key = (f, r.get("Dispatch_Id"))
            disp[key][r["Counter_Name"]] += float(r["Counter_Value"])
            dur = None
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            meta[key] = (r["Kernel_Name"][:60], r.get("Grid_Size"), dur)
print("columns:", cols)
groups = defaultdict(list)
for key, c in disp.items():
    groups[(meta[key][0], meta[key][1])].append((c, meta[key][2]))
for (name, grid), lst in sorted(groups.items()):
    durs = [x[1] for x in lst if x[1] is not None]
    print(f"-- {name} grid {grid}: {len(lst)} dispatch-passes" + (f", mean duration {sum(durs) / len(durs):.1f} us" if durs else ""))
    names = sorted({k for c, _ in lst for k in c})
    for k in names:
        v = [c[k] for c, _ in lst if k in c]
        print(f"   {k:30s} {sum(v) / len(v):18.1f}  (n={len(v)})")
