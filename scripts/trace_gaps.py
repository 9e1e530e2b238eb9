"""Idle time of the device inside a bench run, from a rocprofv3 --kernel-trace CSV (diagnostic).
usage: python scripts/trace_gaps.py <dir with *_kernel_trace.csv> [min_gap_us]
Cuts the trace into segments at gaps >= 2 ms (model set-up, host synchronisations), prints span / union busy time / idle of every
segment longer than 50 ms, and the largest gaps inside the longest one (the timed steps run back to back in one segment)."""
import csv
import glob
import sys

d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
rows.sort()
segs, cur = [], [rows[0]]
end = rows[0][1]
for r in rows[1:]:
    if r[0] - end >= 2_000_000:
        segs.append(cur)
        cur = []
    cur.append(r)
    end = max(end, r[1])
segs.append(cur)


def analyse(seg):
    busy, gaps = 0, []
    cs, ce, last = seg[0][0], seg[0][1], seg[0][2]
    for s, e, n in seg[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append((s - ce, last, n, ce - seg[0][0]))
            cs, ce = s, e
        if e >= ce:
            ce, last = e, n
    busy += ce - cs
    return ce - seg[0][0], busy, gaps


best = None
for seg in segs:
    span, busy, gaps = analyse(seg)
    if span < 50e6:
        continue
    print(f"segment at +{(seg[0][0] - rows[0][0]) / 1e6:9.1f} ms: span {span / 1e6:8.2f} ms, busy {busy / 1e6:8.2f} ms, idle {(span - busy) / 1e6:6.2f} ms "
          f"({100.0 * (span - busy) / span:.2f} %), {len(seg)} kernels")
    if best is None or span > best[0]:
        best = (span, busy, gaps)
if best:
    big = sorted(g for g in best[2] if g[0] >= min_gap * 1e3)[::-1]
    print(f"longest segment: {len(best[2])} gaps, {len(big)} of them >= {min_gap} us (total {sum(g[0] for g in big) / 1e6:.2f} ms)")
    for g in big[:30]:
        print(f"  {g[0] / 1e3:8.1f} us at +{g[3] / 1e6:8.2f} ms  after {g[1]}  before {g[2]}")
