"""Checkpoint layout: tensor-name enumeration, synthetic weights, diffusers-directory IO.

The reference reads/writes these layouts and the build keeps them unchanged
(SURVEY.md 8b "Checkpoint layout", Appendix B):
  <dir>/config.json + <dir>/diffusion_pytorch_model.safetensors   (UNet2DConditionModel.save_pretrained,
                                                                   /root/reference/ldiffusion.py:273; from_pretrained segmentor.py:79)
  <dir>/proj_weights.pt = {weight[768,768], bias[768]}            (ldiffusion.py:274-277; segmentor.py:45-48)
  <sd_dir>/{unet,vae}/...                                         (StableDiffusionImg2ImgPipeline.from_pretrained, ldiffusion.py:67)

There is no network and no SD-v1.5 checkpoint in this environment, so bench and
tests use seeded synthetic weights written in exactly this key layout; a real
checkpoint directory loads through the same code path.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch

WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"


# ------------------------------------------------------------------------------------------
# tensor-name / shape enumeration (diffusers key layout)
# ------------------------------------------------------------------------------------------
def _conv(s, p, cin, cout, k):
    s[p + ".weight"] = (cout, cin, k, k)
    s[p + ".bias"] = (cout,)


def _lin(s, p, cin, cout, bias=True):
    s[p + ".weight"] = (cout, cin)
    if bias:
        s[p + ".bias"] = (cout,)


def _norm(s, p, c):
    s[p + ".weight"] = (c,)
    s[p + ".bias"] = (c,)


def _resnet(s, p, cin, cout, temb_dim):
    _norm(s, p + ".norm1", cin)
    _conv(s, p + ".conv1", cin, cout, 3)
    if temb_dim:
        _lin(s, p + ".time_emb_proj", temb_dim, cout)
    _norm(s, p + ".norm2", cout)
    _conv(s, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(s, p + ".conv_shortcut", cin, cout, 1)


def _transformer(s, p, c, ctx_dim):
    _norm(s, p + ".norm", c)
    _conv(s, p + ".proj_in", c, c, 1)
    b = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        _norm(s, f"{b}.{n}", c)
    for a, kv in (("attn1", c), ("attn2", ctx_dim)):
        _lin(s, f"{b}.{a}.to_q", c, c, bias=False)
        _lin(s, f"{b}.{a}.to_k", kv, c, bias=False)
        _lin(s, f"{b}.{a}.to_v", kv, c, bias=False)
        _lin(s, f"{b}.{a}.to_out.0", c, c)
    _lin(s, f"{b}.ff.net.0.proj", c, 8 * c)
    _lin(s, f"{b}.ff.net.2", 4 * c, c)
    _conv(s, p + ".proj_out", c, c, 1)


def unet_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    lpb = cfg["layers_per_block"]
    _conv(s, "conv_in", cfg["in_channels"], boc[0], 3)
    _lin(s, "time_embedding.linear_1", boc[0], temb)
    _lin(s, "time_embedding.linear_2", temb, temb)
    skip_ch = [boc[0]]
    ch = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            _resnet(s, f"down_blocks.{i}.resnets.{j}", ch, boc[i], temb)
            ch = boc[i]
            if t == "CrossAttnDownBlock2D":
                _transformer(s, f"down_blocks.{i}.attentions.{j}", ch, ctx)
            skip_ch.append(ch)
        if i != len(boc) - 1:
            _conv(s, f"down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
            skip_ch.append(ch)
    _resnet(s, "mid_block.resnets.0", ch, ch, temb)
    _transformer(s, "mid_block.attentions.0", ch, ctx)
    _resnet(s, "mid_block.resnets.1", ch, ch, temb)
    rev = list(reversed(boc))
    for i, t in enumerate(cfg["up_block_types"]):
        for j in range(lpb + 1):
            _resnet(s, f"up_blocks.{i}.resnets.{j}", ch + skip_ch.pop(), rev[i], temb)
            ch = rev[i]
            if t == "CrossAttnUpBlock2D":
                _transformer(s, f"up_blocks.{i}.attentions.{j}", ch, ctx)
        if i != len(boc) - 1:
            _conv(s, f"up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    _norm(s, "conv_norm_out", ch)
    _conv(s, "conv_out", ch, cfg["out_channels"], 3)
    return s


def _vae_mid(s, p, c):
    _resnet(s, p + ".resnets.0", c, c, 0)
    a = p + ".attentions.0"
    _norm(s, a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(s, f"{a}.{n}", c, c)
    _resnet(s, p + ".resnets.1", c, c, 0)


def vae_param_shapes(cfg: dict) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    boc, lpb, lat = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    _conv(s, "encoder.conv_in", cfg["in_channels"], boc[0], 3)
    ch = boc[0]
    for i in range(len(boc)):
        for j in range(lpb):
            _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}", ch, boc[i], 0)
            ch = boc[i]
        if i != len(boc) - 1:
            _conv(s, f"encoder.down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
    _vae_mid(s, "encoder.mid_block", ch)
    _norm(s, "encoder.conv_norm_out", ch)
    _conv(s, "encoder.conv_out", ch, 2 * lat, 3)
    _conv(s, "quant_conv", 2 * lat, 2 * lat, 1)
    _conv(s, "post_quant_conv", lat, lat, 1)
    rev = list(reversed(boc))
    _conv(s, "decoder.conv_in", lat, rev[0], 3)
    _vae_mid(s, "decoder.mid_block", rev[0])
    ch = rev[0]
    for i in range(len(boc)):
        for j in range(lpb + 1):
            _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}", ch, rev[i], 0)
            ch = rev[i]
        if i != len(boc) - 1:
            _conv(s, f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    _norm(s, "decoder.conv_norm_out", ch)
    _conv(s, "decoder.conv_out", ch, cfg["out_channels"], 3)
    return s


def param_count(shapes) -> int:
    n = 0
    for shp in shapes.values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n


# ------------------------------------------------------------------------------------------
# seeded synthetic weights
# ------------------------------------------------------------------------------------------
def synthetic_state_dict(shapes, seed: int, dtype=torch.float32, fp16_values: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic (CPU generator) stand-in weights: conv/linear ~ N(0, 1/fan_in), biases ~ 0.02 N,
    norm gamma ~ 1 + 0.1 N, beta ~ 0.05 N.  Keeps activations O(1) through the ~480-op graph.
    fp16_values=True: every value is fp16-representable (an fp16 checkpoint, BASELINE.json configs[1] "fp16 SD-v1.5 UNet", loaded
    with torch_dtype=float32 as the reference does, ldiffusion.py:67): the HIP path, which stores conv / linear weights in fp16,
    and an fp32 consumer of the same dict then hold identical parameters."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = OrderedDict()
    for name, shp in shapes.items():
        is_norm = ".norm" in name or "group_norm" in name or "conv_norm_out" in name
        if name.endswith(".weight") and is_norm:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias") and is_norm:
            t = 0.05 * torch.randn(shp, generator=g)
        elif name.endswith(".weight"):
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            t = torch.randn(shp, generator=g) * (1.0 / fan_in) ** 0.5
        else:
            t = 0.02 * torch.randn(shp, generator=g)
        if fp16_values:
            t = t.to(torch.float16).to(torch.float32)
        sd[name] = t.to(dtype)
    return sd


# ------------------------------------------------------------------------------------------
# diffusers-directory IO
# ------------------------------------------------------------------------------------------
def save_model_dir(path: str, cfg: dict, sd) -> None:
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, WEIGHTS_NAME))


def load_model_dir(path: str):
    """Returns (config dict, state dict) from a diffusers model directory (safetensors or .bin)."""
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    st = os.path.join(path, WEIGHTS_NAME)
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        binp = os.path.join(path, "diffusion_pytorch_model.bin")
        if not os.path.exists(binp):
            raise FileNotFoundError(f"no {WEIGHTS_NAME} or diffusion_pytorch_model.bin in {path}")
        sd = torch.load(binp, map_location="cpu", weights_only=True)
    return cfg, sd


_VAE_ATTN_RENAMES = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def normalize_vae_keys(sd):
    """Accept the deprecated SD-v1.5 VAE attention names (diffusers remaps them at load)."""
    out = OrderedDict()
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts and len(parts) >= 2 and parts[-2] in _VAE_ATTN_RENAMES:
            parts[-2] = _VAE_ATTN_RENAMES[parts[-2]]
            k = ".".join(parts)
        if v.dim() == 4 and "attentions" in k and k.endswith(".weight") and v.shape[-1] == 1 and "group_norm" not in k:
            v = v[:, :, 0, 0]  # very old checkpoints store the attention linears as 1x1 convs
        out[k] = v
    return out
