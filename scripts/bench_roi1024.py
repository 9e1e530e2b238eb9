"""SURVEY 8f row 1 measurement (not the bench line): the reference-faithful single-pass form on 1024x1024 ROIs at SD-v1.5 width
(128x128 latents, 16,384 tokens in the level-0 self-attention), B ROIs per call, synthetic weights.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib, configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda:0"
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43), dev),
                                      UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42), dev))
s = LaplaceSampler(pipe)
x = torch.rand((B, 3, 1024, 1024), generator=torch.Generator().manual_seed(1)).to(dev)
ctx = (torch.randn((1, 6, 768), generator=torch.Generator().manual_seed(2)) * 0.5).to(dev)
for _ in range(2):
    s.sample(x, ctx, 1, want_features=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    s.sample(x, ctx, 1, want_features=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
lib = _lib.load()
s.set_overlap(0); lib.ldiff_prof_set_filter(None); lib.ldiff_prof_enable(1)
s.sample(x, ctx, 1, want_features=False)
torch.cuda.synchronize()
lib.ldiff_prof_enable(0)
rows = sorted(_lib.prof_collect(), key=lambda r: -r["ms"])[:6]
print(json.dumps({"workload": f"{B} ROIs of 1024x1024, one pass (encode, UNet at 128x128 latents, PLMS, decode to uint8)", "rois_per_sec": B / dt,
                  "ms_per_call": 1e3 * dt, "top_kernels": [{"name": r["name"], "launches": r["launches"], "ms": round(r["ms"], 3),
                                                            "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1)} for r in rows]}))
