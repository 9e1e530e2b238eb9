"""Oracle: duck-typed `pipeline / unet / vae` objects + the sampler loops.
TEST INFRASTRUCTURE ONLY.

The objects expose exactly the attribute surface the reference touches
 so that (a) the reference's own loop bodies can be driven with
them in this container to pin call order (scripts/gen_golden_reference.py) and
(b) the product's shims can be compared against them call by call.

Sampler variants restated here (all file:line into /root/reference):
  sample_v6            pixel_latent_vector.py:58-102   N-pass PLMS, decode every pass, luma features
  sample_one_pass      segmentor.py:86-112, 490-545    set_timesteps(1), one pass at t=1, decoded RGB
  laplace_features_v5  ldiffusion.py:227-247           Laplace forward noise, no scheduler.step, float luma
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import noise_post, schedule, unet as _unet, vae as _vae


class _Cfg:
    def __init__(self, d):
        self.__dict__.update(d)


class OracleUNet:
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.config = _Cfg(cfg)
        self.calls = []  # (timestep:int) log, for call-order pinning

    def __call__(self, sample, timestep, encoder_hidden_states, *args, **kw):
        self.calls.append(int(timestep))
        with torch.no_grad():
            return _unet.unet_forward(self.sd, self.cfg, sample, timestep, encoder_hidden_states, **kw)

    def eval(self):
        return self

    def to(self, *a, **k):
        return self


class _EncOut:
    def __init__(self, moments):
        self.latent_dist = _vae.LatentDist(moments)


class _DecOut:
    def __init__(self, sample):
        self.sample = sample


class OracleVAE:
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg
        self.config = _Cfg(dict(cfg, scaling_factor=cfg.get("scaling_factor", noise_post.SCALING_FACTOR)))

    def encode(self, x):
        with torch.no_grad():
            return _EncOut(_vae.vae_encode_moments(self.sd, self.cfg, x))

    def decode(self, z):
        with torch.no_grad():
            return _DecOut(_vae.vae_decode(self.sd, self.cfg, z))

    def eval(self):
        return self

    def to(self, *a, **k):
        return self


class OraclePipeline:
    """StableDiffusionImg2ImgPipeline stand-in: vae, unet, scheduler, decode_latents, numpy_to_pil."""

    def __init__(self, unet: OracleUNet, vae: OracleVAE, tokenizer=None, text_encoder=None):
        self.unet, self.vae = unet, vae
        self.scheduler = schedule.PNDMOracle()
        self.tokenizer, self.text_encoder = tokenizer, text_encoder

    def decode_latents(self, latents):
        latents = 1 / self.vae.config.scaling_factor * latents
        image = self.vae.decode(latents).sample
        return noise_post.decode_post(image)

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        return [Image.fromarray(im) for im in noise_post.to_uint8(images)]

    def to(self, *a, **k):
        return self


def sample_v6(pipe: OraclePipeline, images: torch.Tensor, ctx: torch.Tensor, num_inference_steps: int):
    """pixel_latent_vector.py:72-86, batched over images.
    Returns dict(latents=[per-pass latents], rgb_u8=[B,N,H,W,3], features=[B,N,H,W] uint8)."""
    latents = pipe.vae.encode(images).latent_dist.mean
    pipe.scheduler.set_timesteps(num_inference_steps - 1)
    lat_hist, rgb, feats = [], [], []
    for t in pipe.scheduler.timesteps:
        latents = pipe.scheduler.scale_model_input(latents, t)
        out = pipe.unet(latents, t, ctx)
        latents = pipe.scheduler.step(out[0], t, latents).prev_sample
        dec = pipe.decode_latents(latents)
        u8 = noise_post.to_uint8(dec)
        lat_hist.append(latents.clone())
        rgb.append(u8)
        feats.append(noise_post.luma_u8(u8))
    return dict(latents=lat_hist, rgb_u8=np.stack(rgb, 1), features=np.stack(feats, 1))


def sample_one_pass(pipe: OraclePipeline, images: torch.Tensor, ctx: torch.Tensor):
    """segmentor.py:99-107 / 519-530: encode mean -> set_timesteps(1) -> one UNet pass -> step -> decode."""
    latents = pipe.vae.encode(images).latent_dist.mean
    pipe.scheduler.set_timesteps(1)
    for t in pipe.scheduler.timesteps:
        latents = pipe.scheduler.scale_model_input(latents, t)
        out = pipe.unet(latents, t, ctx)
        latents = pipe.scheduler.step(out[0], t, latents).prev_sample
        dec = pipe.decode_latents(latents)
    return dict(latents=latents, rgb_u8=noise_post.to_uint8(dec))


def laplace_features_v5(pipe: OraclePipeline, images: torch.Tensor, ctx: torch.Tensor, n_sched: int, u_list, out_hw=64):
    """ldiffusion.py:228-247 (forward part): z0 fixed; per step x_t = z0 + Laplace(0, sqrt(1-abar_t));
    eps = unet(x_t); rgb = bilinear(vae.decode(eps).sample, 64x64); gray = luma_float(rgb); concat over steps.
    `u_list[i]` is the uniform draw for step i (parity is defined given u)."""
    z0 = pipe.vae.encode(images).latent_dist.mean
    pipe.scheduler.set_timesteps(n_sched)
    grays, rgb = [], None
    for i, t in enumerate(pipe.scheduler.timesteps):
        lat = pipe.scheduler.scale_model_input(z0, t)
        noisy = noise_post.laplace_forward_noise(lat, pipe.scheduler.alphas_cumprod[int(t)], u_list[i])
        den = pipe.unet(noisy, t, ctx).sample
        rgb = F.interpolate(pipe.vae.decode(den).sample, size=(out_hw, out_hw), mode="bilinear", align_corners=False)
        grays.append(noise_post.luma_float(rgb))
    return dict(gray=torch.cat(grays, dim=1), rgb=rgb)
