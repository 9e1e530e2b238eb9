"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, separate runs) over the bench workload
into profiles/pmc_traffic_<kernel source hash>.json: per-dispatch HBM traffic of each contraction kernel, keyed by bench.py's row names.
usage: python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    a = acc[r["Kernel_Name"]]
                    a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


def bench_name(k):
    m = re.search(r"conv3x3w_kernel<(\d+), (true|false)>", k)
    if m: return f"conv3x3<8x16,{m.group(1)}{',gn' if m.group(2) == 'true' else ''}>"
    if re.search(r"conv3x3d_kernel<\d+>", k): return "conv3x3<16x16d,128,gn>"   # producer / consumer kernel: every epilogue configuration
    m = re.search(r"conv3x3p_kernel<(\d+), (true|false), (\d+)>", k)   # persistent 16x16 kernel: every epilogue configuration / parity mode of a tile width
    if m: return f"conv3x3<16x16,{m.group(1)}>"
    m = re.search(r"conv3x3_kernel<(\d+), (\d+), (\d+), (true|false)>", k)
    if m: return f"conv3x3<{m.group(1)}x{m.group(2)},{m.group(3)}{',gn' if m.group(4) == 'true' else ''}>"
    m = re.search(r"gemm_dma_kernel<(\d+), (\d+)>", k)
    if m: return f"gemm_dma<{m.group(1)},{m.group(2)}>"
    m = re.search(r"lngemm_kernel<(\d+), (true|false),", k)
    if m: return f"lngemm<{m.group(1)}{',geglu' if m.group(2) == 'true' else ''}>"
    if "attn_dsplit_kernel" in k: return "attn<512,512>"
    m = re.search(r"attn_kernel<(\d+), (\d+),", k)
    if m: return f"attn<{m.group(1)},{m.group(2)}>"
    return None


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python3 bench.py --steps 1 "
                "--warmup 0 --no-cpu-baseline --no-unet-step --no-prof` on MI355X; per-dispatch averages in KB. traffic_bytes = "
                "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 reports half the bytes of 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM); "
                "the factor was calibrated on the conv3x3 kernel with a known byte count (128->128 at 8x512x512: FETCH_SIZE 264.5 MB "
                "for 537 MB of input).", "kernels": {}}
agg = defaultdict(lambda: [0.0, 0, 0.0, 0])   # several kernel instantiations can share one bench row
for k, (fs, n) in fetch.items():
    name = bench_name(k)
    if not name or k not in write or n == 0: continue
    ws, wn = write[k]
    a = agg[name]
    a[0] += fs; a[1] += n; a[2] += ws; a[3] += wn
for name, (fs, n, ws, wn) in agg.items():
    f_kb, w_kb = fs / n, ws / max(wn, 1)
    out["kernels"][name] = dict(dispatches=n, fetch_kb=round(f_kb, 1), write_kb=round(w_kb, 1), traffic_bytes=round((2 * f_kb + w_kb) * 1024))
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["dispatches"])[:8]: print(k, v)
