"""One eager SD-v1.5 UNet pass (B = LDIFF_UNET_B, default 8; 64x64 latents, default precision) with HIP events on every contraction / norm launch, one line per
launch (LDIFF_PROF_DUMP): which launches are long AND slow?  Diagnostic; prints the launches sorted by time."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("LDIFF_PROF_DUMP") is None:
    env = dict(os.environ, LDIFF_PROF_DUMP="1")
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    rows = []
    for l in out.stderr.splitlines():
        m = re.match(r"\[ldiff_prof\] (\S+)\s+([\d.]+) us\s+([\d.]+) GFLOP\s+([\d.]+) TFLOP/s\s+([\d.]+) MB\s+([\d.]+) GB/s", l)
        if m:
            rows.append((m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6))))
    tot = sum(r[1] for r in rows)
    print(f"{len(rows)} profiled launches, {tot / 1e3:.2f} ms")
    # launches of one kernel with the same work are one shape: count, total time, rate
    groups = {}
    for r in rows:
        g = groups.setdefault((r[0], round(r[2], 2), round(r[4], 1)), [0, 0.0])
        g[0] += 1; g[1] += r[1]
    print(f"{'kernel':28s} {'GFLOP':>9s} {'MB':>8s} {'n':>3s} {'total us':>9s} {'us each':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
    for (name, gf, mb), (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{name:28s} {gf:9.2f} {mb:8.1f} {n:3d} {us:9.1f} {us / n:8.1f} {gf * n / us * 1e-3 * 1e3:8.1f} {mb * n / us * 1e3 * 1e-3:7.1f}")
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from ldiffusion_amd import _lib, configs, weights
from ldiffusion_amd.models import UNet2DConditionModel
ucfg = configs.SD15_UNET
unet = UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), "cuda:0")
unet.set_graph(False)
lat = torch.randn((int(os.environ.get("LDIFF_UNET_B", "8")), 4, 64, 64), device="cuda:0")   # LDIFF_UNET_B: batch (default 8; 1 = the reference's own batch)
ctx = torch.randn((1, 6, 768), device="cuda:0") * 0.5
for _ in range(2):
    unet(lat, 501, ctx)
torch.cuda.synchronize()
lib = _lib.load()
lib.ldiff_prof_set_filter(None)
lib.ldiff_prof_enable(1)
unet(lat, 501, ctx)
torch.cuda.synchronize()
lib.ldiff_prof_enable(0)
_lib.prof_collect()
