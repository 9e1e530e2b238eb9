"""Oracle: AutoencoderKL encode/decode, restated in plain torch fp32 (CPU).
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (diffusers absent; see oracle/__init__).

Mirrors diffusers 0.34.0 `AutoencoderKL` (Encoder / Decoder / UNetMidBlock2D /
DownEncoderBlock2D / UpDecoderBlock2D) for the SD-v1.5 vae/config.json
(SURVEY.md 8a R5).  Consumes a diffusers-layout state dict; accepts both the
current attention key names (to_q/to_k/to_v/to_out.0) and the deprecated ones
the original SD-v1.5 file uses (query/key/value/proj_attn).

Reference call sites:
  vae.encode(x).latent_dist.mean      segmentor.py:99,437,519  pixel_latent_vector.py:73  ldiffusion.py:228  utils.py:190
  vae.encode(x).latent_dist.sample()  segmentor.py:339
  vae.decode(z).sample                ldiffusion.py:240  segmentor.py:379  (and inside decode_latents)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .unet import _conv, _gn, _lin, resnet_block

VAE_EPS = 1e-6


def _attn_key(sd, p, new, old):
    return p + "." + (new if (p + "." + new + ".weight") in sd else old)


def vae_mid_attention(sd, p, x, groups):
    """Single-head `Attention(residual_connection=True, norm_num_groups=32, bias=True)` of UNetMidBlock2D."""
    B, C, H, W = x.shape
    gn = p + ".group_norm"
    h = F.group_norm(x.view(B, C, H * W), groups, sd[gn + ".weight"], sd[gn + ".bias"], VAE_EPS)
    h = h.transpose(1, 2)  # [B, HW, C]
    q = _lin(sd, _attn_key(sd, p, "to_q", "query"), h)
    k = _lin(sd, _attn_key(sd, p, "to_k", "key"), h)
    v = _lin(sd, _attn_key(sd, p, "to_v", "value"), h)
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = _lin(sd, _attn_key(sd, p, "to_out.0", "proj_attn"), o)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def _mid(sd, p, h, groups):
    h = resnet_block(sd, p + ".resnets.0", h, None, groups, VAE_EPS)
    h = vae_mid_attention(sd, p + ".attentions.0", h, groups)
    return resnet_block(sd, p + ".resnets.1", h, None, groups, VAE_EPS)


def vae_encode_moments(sd, cfg, x):
    """Encoder + quant_conv -> moments [B, 2*latent, h, w] (mean | logvar)."""
    x = x.to(torch.float32)
    groups, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    h = _conv(sd, "encoder.conv_in", x)
    for i in range(len(boc)):
        for j in range(lpb):
            h = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, groups, VAE_EPS)
        if i != len(boc) - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)  # Downsample2D(padding=0): asymmetric
            h = _conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0)
    h = _mid(sd, "encoder.mid_block", h, groups)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, groups, VAE_EPS))
    h = _conv(sd, "encoder.conv_out", h)
    return _conv(sd, "quant_conv", h, padding=0)


class LatentDist:
    """DiagonalGaussianDistribution: `.mean`, `.sample()`."""

    def __init__(self, moments):
        self.mean, self.logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * noise


def vae_decode(sd, cfg, z):
    """post_quant_conv + Decoder -> [B, 3, 8h, 8w]."""
    z = z.to(torch.float32)
    groups, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    h = _conv(sd, "post_quant_conv", z, padding=0)
    h = _conv(sd, "decoder.conv_in", h)
    h = _mid(sd, "decoder.mid_block", h, groups)
    for i in range(len(boc)):
        for j in range(lpb + 1):
            h = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, groups, VAE_EPS)
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, groups, VAE_EPS))
    return _conv(sd, "decoder.conv_out", h)
