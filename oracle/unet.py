"""Oracle: UNet2DConditionModel forward, restated in plain torch fp32 (CPU).
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (diffusers absent; see oracle/__init__).

Mirrors diffusers 0.34.0 `UNet2DConditionModel` for SD-v1.5-style configs
(SURVEY.md 8a R1-R4, Appendix B).  Consumes a state dict in the diffusers key
layout (`diffusion_pytorch_model.safetensors`) and a `config.json`-style dict.

Reference call sites: `unet(latents, t, text_embeddings)` -> `[0]` / `.sample`
  segmentor.py:103,444,526   pixel_latent_vector.py:78   ldiffusion.py:160,238   utils.py:201
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


class UNetOutput:
    """Has both `.sample` and `[0]` like diffusers' UNet2DConditionOutput."""

    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


def timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool, freq_shift: float) -> torch.Tensor:
    """diffusers.models.embeddings.get_timestep_embedding (R2)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(0, half, dtype=torch.float32)
    exponent = exponent / (half - freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet_block(sd, p, x, temb, groups, eps):
    """ResnetBlock2D (R3).  `temb` is the SiLU-free time embedding or None (VAE)."""
    h = F.silu(_gn(sd, p + ".norm1", x, groups, eps))
    h = _conv(sd, p + ".conv1", h)
    if temb is not None:
        h = h + _lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(_gn(sd, p + ".norm2", h, groups, eps))
    h = _conv(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd, p, x, ctx, heads):
    """diffusers Attention + AttnProcessor2_0 (no mask): q from x, k/v from ctx (or x)."""
    ctx = x if ctx is None else ctx
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx)
    B, Lq, C = q.shape
    d = C // heads
    q = q.view(B, Lq, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)  # scale 1/sqrt(d), softmax fp32
    o = o.transpose(1, 2).reshape(B, Lq, C)
    return _lin(sd, p + ".to_out.0", o)


def transformer_block(sd, p, x, ctx, heads):
    """BasicTransformerBlock (R4): LN->self-attn, LN->cross-attn, LN->GEGLU FF, each + residual."""
    C = x.shape[-1]
    ln = lambda n, t: F.layer_norm(t, (C,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)
    x = attention(sd, p + ".attn1", ln("norm1", x), None, heads) + x
    x = attention(sd, p + ".attn2", ln("norm2", x), ctx, heads) + x
    h = _lin(sd, p + ".ff.net.0.proj", ln("norm3", x))
    h, gate = h.chunk(2, dim=-1)
    h = h * F.gelu(gate)  # erf GELU
    return _lin(sd, p + ".ff.net.2", h) + x


def transformer2d(sd, p, x, ctx, heads, groups):
    """Transformer2DModel, use_linear_projection=False (R4): GN(eps 1e-6) -> conv1x1 -> blocks -> conv1x1 + res."""
    B, C, H, W = x.shape
    res = x
    h = _gn(sd, p + ".norm", x, groups, 1e-6)
    h = _conv(sd, p + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = transformer_block(sd, p + ".transformer_blocks.0", h, ctx, heads)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    h = _conv(sd, p + ".proj_out", h, padding=0)
    return h + res


def unet_forward(sd, cfg, sample, timestep, ctx, down_block_additional_residuals=None,
                 mid_block_additional_residual=None) -> UNetOutput:
    """UNet2DConditionModel.forward for CrossAttnDown/Down/UpBlock/CrossAttnUp stacks."""
    sample = sample.to(torch.float32)
    ctx = ctx.to(torch.float32)
    B = sample.shape[0]
    if ctx.shape[0] != B:
        ctx = ctx.expand(B, -1, -1)
    boc = cfg["block_out_channels"]
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    heads = cfg["attention_head_dim"]  # SD-v1.5 quirk: this field is the number of heads
    lpb = cfg["layers_per_block"]

    t = torch.as_tensor(timestep)
    if t.dim() == 0:
        t = t[None]
    t = t.expand(B)
    temb = timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", temb)))

    h = _conv(sd, "conv_in", sample)
    skips = [h]
    for i, btype in enumerate(cfg["down_block_types"]):
        has_attn = btype == "CrossAttnDownBlock2D"
        for j in range(lpb):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, temb, groups, eps)
            if has_attn:
                h = transformer2d(sd, f"down_blocks.{i}.attentions.{j}", h, ctx, heads, groups)
            skips.append(h)
        if i != len(boc) - 1:
            h = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    if down_block_additional_residuals is not None:  # ControlNet residuals (segmentor.py:366-372)
        skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]

    h = resnet_block(sd, "mid_block.resnets.0", h, temb, groups, eps)
    h = transformer2d(sd, "mid_block.attentions.0", h, ctx, heads, groups)
    h = resnet_block(sd, "mid_block.resnets.1", h, temb, groups, eps)
    if mid_block_additional_residual is not None:
        h = h + mid_block_additional_residual

    for i, btype in enumerate(cfg["up_block_types"]):
        has_attn = btype == "CrossAttnUpBlock2D"
        for j in range(lpb + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, temb, groups, eps)
            if has_attn:
                h = transformer2d(sd, f"up_blocks.{i}.attentions.{j}", h, ctx, heads, groups)
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h)

    h = F.silu(_gn(sd, "conv_norm_out", h, groups, eps))
    return UNetOutput(_conv(sd, "conv_out", h))
