"""Stress / race check (diagnostic, GPU): many sampler calls with varying batch, size and pass count, decode side stream on
(join per call and deferred join with two pipelines) against the single-stream result, bit for bit; device memory must not grow."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline

DEV = "cuda:0"
ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43), DEV),
                                      UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42), DEV))
ser, ov, d1, d2 = (LaplaceSampler(pipe) for _ in range(4))
ser.set_overlap(0); d1.set_overlap(2); d2.set_overlap(2)
rng = random.Random(0)
g = torch.Generator().manual_seed(0)
ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
free0 = None
pending = None
n_checked = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    B, H, W, N = rng.choice([1, 2, 3, 5]), rng.choice([64, 128]), rng.choice([64, 128, 192]), rng.choice([1, 3, 4, 5, 7])
    x = torch.rand((B, 3, H, W), generator=g).to(DEV)
    ref = ser.sample(x, ctx, N)
    a = ov.sample(x, ctx, N)
    cur = ((d1, d2)[it % 2], (d1, d2)[it % 2].sample(x, ctx, N), ref)
    for k in ("latents", "features", "rgb"):
        assert torch.equal(a[k], ref[k]), (it, k, "join-per-call")
    if pending is not None:
        pending[0].join()
        for k in ("latents", "features", "rgb"):
            assert torch.equal(pending[1][k], pending[2][k]), (it, k, "deferred join")
        n_checked += 1
    pending = cur
    if it == 9:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
pending[0].join()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print(f"ok: {n_checked + 1} deferred-join and {it + 1} join-per-call results identical to the single-stream sampler; "
      f"free device memory after 10 calls {free0 >> 20} MiB, at the end {free1 >> 20} MiB")
assert free0 is None or free0 - free1 < (256 << 20), "device memory keeps growing"
