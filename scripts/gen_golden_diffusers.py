"""Pin the oracle's diffusers arithmetic to REAL diffusers -- to be run wherever `diffusers==0.34.0` is importable.

PARITY STATUS: the UNet2DConditionModel / AutoencoderKL / PNDMScheduler / decode_latents restatements under oracle/ are UNPINNED in
this repository: diffusers is not under /root/reference, not installed in the build container or on the GPU box, and there is no
network.  This script closes that gap the moment a diffusers checkout or wheel is reachable:

    DIFFUSERS_SRC=/path/to/diffusers/src python scripts/gen_golden_diffusers.py        # or with diffusers installed: no variable
    SD15_DIR=/path/to/stable-diffusion-v1-5 python scripts/gen_golden_diffusers.py     # additionally pins a real-checkpoint vector

It instantiates the real classes with this repo's reduced-width configs (ldiffusion_amd/configs.py TINY_*), loads the same seeded
synthetic state dicts STRICTLY (which also pins the tensor-name layout of ldiffusion_amd/weights.py), runs them on seeded inputs and
writes tests/golden/diffusers_tiny.npz (inputs + outputs only).  tests/test_cpu_oracle.py::test_oracle_against_real_diffusers_when_pinned
then checks oracle/ against that file and the "parity unpinned" notes in oracle/__init__.py, DESIGN.md and README.md can be dropped.
Without diffusers the script exits with status 2 and writes nothing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("DIFFUSERS_SRC"):
    sys.path.insert(0, os.environ["DIFFUSERS_SRC"])
try:
    import diffusers
    from diffusers import AutoencoderKL, PNDMScheduler, UNet2DConditionModel
except Exception as e:  # noqa: BLE001
    print(f"diffusers is not importable here ({e!r}); set DIFFUSERS_SRC or install diffusers==0.34.0.  Nothing written.")
    sys.exit(2)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ldiffusion_amd import configs, weights  # noqa: E402

out = {"diffusers_version": np.array(diffusers.__version__)}
torch.manual_seed(0)
g = torch.Generator().manual_seed(2024)

# ---- UNet2DConditionModel ----
ucfg = {k: v for k, v in configs.TINY_UNET.items() if not k.startswith("_")}
unet = UNet2DConditionModel(**ucfg).eval()
usd = weights.synthetic_state_dict(weights.unet_param_shapes(configs.TINY_UNET), 42)
unet.load_state_dict(usd, strict=True)
x = torch.randn((2, 4, 16, 16), generator=g)
ctx = torch.randn((2, 6, ucfg["cross_attention_dim"]), generator=g) * 0.5
with torch.no_grad():
    for t in (1, 501, 751):
        out[f"unet_t{t}"] = unet(x, torch.tensor(t), ctx).sample.numpy()
out["unet_x"], out["unet_ctx"] = x.numpy(), ctx.numpy()

# ---- AutoencoderKL (R5) ----
vcfg = {k: v for k, v in configs.TINY_VAE.items() if not k.startswith("_")}
vae = AutoencoderKL(**vcfg).eval()
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(configs.TINY_VAE), 43)
vae.load_state_dict(vsd, strict=True)
img = torch.rand((2, 3, 64, 64), generator=g)
z = torch.randn((2, 4, 8, 8), generator=g) * 0.5
with torch.no_grad():
    dist = vae.encode(img).latent_dist
    out["vae_img"], out["vae_mean"], out["vae_logvar"] = img.numpy(), dist.mean.numpy(), dist.logvar.numpy()
    out["vae_z"], out["vae_dec"] = z.numpy(), vae.decode(z).sample.numpy()

# ---- PNDMScheduler with the SD-v1.5 scheduler_config.json values (R6) ----
sch = PNDMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True,
                    set_alpha_to_one=False, steps_offset=1)
out["alphas_cumprod"] = sch.alphas_cumprod.numpy()
for n in (1, 4, 9, 19):
    sch.set_timesteps(n)
    out[f"timesteps_{n}"] = sch.timesteps.numpy()
    xs = torch.randn((1, 4, 8, 8), generator=g)
    seq = [xs.numpy()]
    eps_all = []
    for t in sch.timesteps:
        eps = torch.randn((1, 4, 8, 8), generator=g)
        eps_all.append(eps.numpy())
        xs = sch.step(eps, t, xs).prev_sample
        seq.append(xs.numpy())
    out[f"plms_{n}_eps"], out[f"plms_{n}_x"] = np.stack(eps_all), np.stack(seq)

# ---- decode_latents / numpy_to_pil (R8) through the real pipeline class, with the tiny VAE ----
try:
    from diffusers import StableDiffusionImg2ImgPipeline
    dl = StableDiffusionImg2ImgPipeline.decode_latents
    class _P:  # the method only touches self.vae
        pass
    p = _P()
    p.vae = vae
    with torch.no_grad():
        out["decode_latents"] = dl(p, z)
    out["numpy_to_pil_u8"] = np.stack([np.asarray(im) for im in StableDiffusionImg2ImgPipeline.numpy_to_pil(out["decode_latents"])])
except Exception as e:  # noqa: BLE001
    print("decode_latents not pinned:", repr(e))

# ---- optional: one real-checkpoint vector ----
sd15 = os.environ.get("SD15_DIR")
if sd15:
    ru = UNet2DConditionModel.from_pretrained(sd15, subfolder="unet", torch_dtype=torch.float32).eval()
    xs = torch.randn((1, 4, 16, 16), generator=g)
    cs = torch.randn((1, 6, 768), generator=g) * 0.5
    with torch.no_grad():
        out["sd15_unet_x"], out["sd15_unet_ctx"], out["sd15_unet_t501"] = xs.numpy(), cs.numpy(), ru(xs, torch.tensor(501), cs).sample.numpy()

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "diffusers_tiny.npz"), **out)
print("wrote tests/golden/diffusers_tiny.npz with", sorted(out))
