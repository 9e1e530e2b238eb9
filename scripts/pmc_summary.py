"""Summarise rocprofv3 --pmc output: mean counter value per dispatch for kernels whose name contains a substring.
usage: python scripts/pmc_summary.py <dir> <kernel-substring>"""
import csv, glob, os, sys
from collections import defaultdict
d, sub = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if sub in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(f"{k:32s} {acc[k][0]/acc[k][1]:16.1f}  (n={acc[k][1]})")
