"""Model configs in the diffusers `config.json` vocabulary.

SD15_UNET / SD15_VAE reproduce the public SD-v1.5 `unet/config.json` and
`vae/config.json` fields the sampling path depends on:
the reference loads exactly these through `from_pretrained`
(/root/reference/segmentor.py:77-80, ldiffusion.py:66-70).
TINY_* are reduced-width graphs with the same topology, used by the parity
tests so the CPU oracle finishes in seconds.
"""
from __future__ import annotations

import copy

SD15_UNET = {
    "_class_name": "UNet2DConditionModel",
    "sample_size": 64,
    "in_channels": 4,
    "out_channels": 4,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "block_out_channels": [320, 640, 1280, 1280],
    "layers_per_block": 2,
    "attention_head_dim": 8,  # SD-v1.5 quirk: number of heads
    "cross_attention_dim": 768,
    "norm_num_groups": 32,
    "norm_eps": 1e-5,
    "act_fn": "silu",
    "flip_sin_to_cos": True,
    "freq_shift": 0,
    "downsample_padding": 1,
    "center_input_sample": False,
    "use_linear_projection": False,
}

SD15_VAE = {
    "_class_name": "AutoencoderKL",
    "in_channels": 3,
    "out_channels": 3,
    "latent_channels": 4,
    "block_out_channels": [128, 256, 512, 512],
    "down_block_types": ["DownEncoderBlock2D"] * 4,
    "up_block_types": ["UpDecoderBlock2D"] * 4,
    "layers_per_block": 2,
    "norm_num_groups": 32,
    "act_fn": "silu",
    "sample_size": 512,
    "scaling_factor": 0.18215,
}

# Same topology, 1/5 width (head dims 8/16/32/32), cross-attention dim 64.
TINY_UNET = dict(copy.deepcopy(SD15_UNET), block_out_channels=[64, 128, 256, 256], cross_attention_dim=64)
# Same topology, 1/4 width.
TINY_VAE = dict(copy.deepcopy(SD15_VAE), block_out_channels=[32, 64, 128, 128])


# diffusers' constructor defaults for the fields the sampling path reads: a config.json written by an older diffusers (or a
# hand-trimmed one) may omit any of them, and `from_pretrained` then fills these in (UNet2DConditionModel.__init__ /
# AutoencoderKL.__init__ signatures of diffusers 0.34.0 [mem]).
UNET_DEFAULTS = {
    "sample_size": None, "in_channels": 4, "out_channels": 4, "center_input_sample": False, "flip_sin_to_cos": True, "freq_shift": 0,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"],
    "block_out_channels": [320, 640, 1280, 1280], "layers_per_block": 2, "downsample_padding": 1, "act_fn": "silu",
    "norm_num_groups": 32, "norm_eps": 1e-5, "cross_attention_dim": 1280, "attention_head_dim": 8, "use_linear_projection": False,
}
VAE_DEFAULTS = {
    "in_channels": 3, "out_channels": 3, "down_block_types": ["DownEncoderBlock2D"], "up_block_types": ["UpDecoderBlock2D"],
    "block_out_channels": [64], "layers_per_block": 1, "act_fn": "silu", "latent_channels": 4, "norm_num_groups": 32, "sample_size": 32,
    "scaling_factor": 0.18215,
}


def with_defaults(cfg: dict, defaults: dict) -> dict:
    out = dict(defaults)
    out.update(cfg)
    return out


def validate_unet_config(cfg: dict) -> None:
    """Reject configs outside what the HIP executor implements (fail loudly, no fallback)."""
    boc = cfg["block_out_channels"]
    n = len(boc)
    if len(cfg["down_block_types"]) != n or len(cfg["up_block_types"]) != n:
        raise ValueError("down/up block lists must match block_out_channels")
    for t in cfg["down_block_types"]:
        if t not in ("CrossAttnDownBlock2D", "DownBlock2D"):
            raise ValueError(f"unsupported down block type {t}")
    for t in cfg["up_block_types"]:
        if t not in ("CrossAttnUpBlock2D", "UpBlock2D"):
            raise ValueError(f"unsupported up block type {t}")
    heads = cfg["attention_head_dim"]
    if not isinstance(heads, int):
        raise ValueError("per-block attention_head_dim lists are not supported")
    for c in boc:
        if c % cfg["norm_num_groups"] or c % 8:
            raise ValueError(f"channel count {c} must be divisible by norm_num_groups and by 8")
        if (c // heads) % 8 or c % heads:
            raise ValueError(f"head dim {c}/{heads} must be a multiple of 8")
    if cfg["cross_attention_dim"] % 8:
        raise ValueError("cross_attention_dim must be a multiple of 8")
    if cfg.get("use_linear_projection", False):
        raise ValueError("use_linear_projection=True is not supported")
    if cfg.get("act_fn", "silu") != "silu":
        raise ValueError("only SiLU is supported")


def validate_vae_config(cfg: dict) -> None:
    for c in cfg["block_out_channels"]:
        if c % cfg["norm_num_groups"] or c % 8:
            raise ValueError(f"channel count {c} must be divisible by norm_num_groups and by 8")
    if cfg["latent_channels"] > 8 or cfg["in_channels"] > 8 or cfg["out_channels"] > 8:
        raise ValueError("in/out/latent channels above 8 are not supported")
