"""Diagnostic: one SD-v1.5 UNet pass (B=8, 64x64 latents): wall time without events vs the sum of per-launch event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib, configs, weights
from ldiffusion_amd.models import UNet2DConditionModel
ucfg = configs.SD15_UNET
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
unet = UNet2DConditionModel(ucfg, usd, "cuda:0")
lat = torch.randn((8, 4, 64, 64), device="cuda:0")
ctx = torch.randn((1, 6, 768), device="cuda:0") * 0.5
lib = _lib.load()
for _ in range(3): unet(lat, 501, ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): unet(lat, 501, ctx)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10 * 1e3
lib.ldiff_prof_set_filter(None); lib.ldiff_prof_enable(1)
unet(lat, 501, ctx)
torch.cuda.synchronize()
lib.ldiff_prof_enable(0)
rows = _lib.prof_collect()
tot = sum(r["ms"] for r in rows); n = sum(r["launches"] for r in rows)
print(f"wall {wall:.2f} ms; profiled launches {n}, sum of event times {tot:.2f} ms")
for r in sorted(rows, key=lambda r: -r["ms"])[:25]:
    print(f"  {r['name']:28s} {r['launches']:4d} {r['ms']:7.3f} ms  avg {1e3*r['ms']/r['launches']:7.1f} us  {r['flops']/(r['ms']*1e-3)/1e12:7.1f} TF")
