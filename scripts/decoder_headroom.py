"""max |activation| per graph stage of the bench configuration (SD-v1.5-size UNet / VAE, 512 x 512, 5 passes, B = 1) against fp16's 65504:
runs the sampler once with LDIFF_TRACE_ABSMAX=1 (include/ldiff.h, "Non-finite detection") and prints, per graph, the stages in order with their
maxima over the run plus the smallest headroom.  The weights here are the seeded synthetic ones of bench.py -- the figures say how the GRAPH
scales activations, not what a trained checkpoint does: trace a real checkpoint the same way before trusting decoder mode 0 with it
(the reference decodes z / 0.18215 of UN-scaled latents, pixel_latent_vector.py:73,81)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("LDIFF_TRACE_ABSMAX") is None:
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, LDIFF_TRACE_ABSMAX="1"), capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    stages, order = {}, []
    for l in out.stderr.splitlines():
        m = re.match(r"\[absmax\] (\S+)\s+(.+?)\s+\[(\d+),(\d+),(\d+),(\d+)\]( split)? max\|x\| = (\S+)", l)
        if m:
            key = (m.group(1), m.group(2).strip())
            if key not in stages:
                order.append(key); stages[key] = [0.0, 0, f"{m.group(4)}x{m.group(5)}x{m.group(6)}"]
            v = float(m.group(8))
            stages[key][0] = max(stages[key][0], v) if v == v else float("nan")
            stages[key][1] += 1
    worst = {}
    for g, st in order:
        mx, n, shp = stages[(g, st)]
        print(f"{g:8s} {st:30s} {shp:14s} max|x| over {n:2d} calls = {mx:10.4g}   headroom x{65504.0 / mx if mx > 0 else float('inf'):8.1f}")
        worst[g] = max(worst.get(g, 0.0), mx)
    for g, mx in worst.items():
        print(f"==> {g}: largest activation {mx:.4g} = 1 / {65504.0 / mx:.1f} of fp16's range")
    sys.exit(out.returncode)
sys.path.insert(0, ROOT)
import torch
from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, "cuda:0"), UNet2DConditionModel(ucfg, usd, "cuda:0"))
pipe.unet.set_graph(False)   # (the trace synchronises per stage: eager launches)
s = LaplaceSampler(pipe)
s.set_overlap(0)
g = torch.Generator().manual_seed(1234)
x = torch.rand((1, 3, 512, 512), generator=g).cuda()
ctx = (torch.randn((1, 6, 768), generator=torch.Generator().manual_seed(1235)) * 0.5).cuda()
out = s.sample(x, ctx, 5)
s.check_finite()
print(f"sampler B=1 512x512 x 5 passes: final latents max|z| = {out['latents'].abs().max().item():.3f}; decoder input z / 0.18215 up to {out['latents'].abs().max().item() / 0.18215:.1f}; check_finite passed")
