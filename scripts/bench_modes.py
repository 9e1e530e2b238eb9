"""Time the three graphs of the sampler (VAE encode, UNet pass, VAE decode) at BASELINE configs[1] size (B=8, 512x512, SD-v1.5 width)
under each storage policy (ldiff_*_set_precision 0/1/2).  usage: python scripts/bench_modes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel

DEV = "cuda:0"
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
unet = UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), DEV)
vae = AutoencoderKL(vcfg, weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True), DEV)
g = torch.Generator().manual_seed(0)
img = torch.rand((8, 3, 512, 512), generator=g).to(DEV)
z = torch.randn((8, 4, 64, 64), generator=g).to(DEV)
ctx = (torch.randn((1, 6, 768), generator=g) * 0.5).to(DEV)


def t(fn, n=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for mode in (0, 1, 2):
    unet.set_precision(mode)
    vae.set_precision(mode, mode)
    print(f"mode {mode}: encode {t(lambda: vae.encode(img)):.2f} ms   unet {t(lambda: unet(z, 501, ctx)):.2f} ms   decode {t(lambda: vae._decode(z, 1 / 0.18215, want_rgb=True)):.2f} ms", flush=True)
