import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from ldiffusion_amd import configs, weights, _lib
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline, argmax_mask
from oracle import noise_post
DEV = "cuda:0"
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43)
pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, DEV), UNet2DConditionModel(ucfg, usd, DEV))
g = torch.Generator().manual_seed(77)
x = torch.rand((1, 3, 64, 64), generator=g).to(DEV)
ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
s = LaplaceSampler(pipe)
for N in (1, 3, 4, 5):
    fused = s.sample(x, ctx, N)["latents"]
    lat = pipe.vae.encode(x).latent_dist.mean
    pipe.scheduler.set_timesteps(1 if N == 1 else N - 1, device=DEV)
    for t in pipe.scheduler.timesteps:
        o = pipe.unet(lat, t, ctx)
        lat = pipe.scheduler.step(o[0], t, lat).prev_sample
    print("N", N, "fused vs stepwise max diff", (fused - lat).abs().max().item())
# determinism of unet
lat = pipe.vae.encode(x).latent_dist.mean
a = pipe.unet(lat, 501, ctx).sample; b = pipe.unet(lat, 501, ctx).sample
print("unet run-to-run diff", (a - b).abs().max().item())
e1 = pipe.vae.encode(x).latent_dist.mean; e2 = pipe.vae.encode(x).latent_dist.mean
print("enc run-to-run diff", (e1 - e2).abs().max().item())
# argmax
g = torch.Generator().manual_seed(11)
logits = torch.randn((2, 6, 33, 47), generator=g)
logits[0, :, 0, 0] = 1.0
logits[0, 3, 0, 1] = float("nan")
m = argmax_mask(logits.to(DEV)).cpu().numpy(); r = noise_post.argmax_mask(logits)
bad = np.argwhere(m != r); print("argmax mismatches", len(bad), bad[:5], [(m[tuple(i)], r[tuple(i)], logits[i[0], :, i[1], i[2]].tolist()) for i in bad[:3]])
