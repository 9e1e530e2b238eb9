"""Mirror of the sampler call site in the reference's utils.py (nnU-Net dataset creation): `copy_or_convert_image`, sampler variant V4
(/root/reference/utils.py:176-208).  Same name, arguments and effect -- a PNG of the one-pass diffusion of the image at `dst_path`, or a plain
copy -- with the arithmetic on the HIP kernels (no CPU fallback: the pipeline shims raise without the library or a GPU)."""
from __future__ import annotations

import shutil

import numpy as np
import torch
from torch import nn

from .pipeline import LaplaceSampler

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
_samplers: dict = {}


def _sampler(pipeline, unet):
    key = (id(pipeline), id(unet))
    s = _samplers.get(key)
    if s is None or s.pipeline is not pipeline or pipeline.unet is not unet:
        pipeline.unet = unet
        s = _samplers[key] = LaplaceSampler(pipeline)
    return s


@torch.no_grad()
def copy_or_convert_image(img, src_path, dst_path, pipeline=None, unet=None, use_diffusion=True):
    """utils.py:176-208.  use_diffusion: Resize((1024, 1024)) + ToTensor + ImageNet Normalize (:180-185) -> vae.encode().mean ->
    set_timesteps(1) -> text embeddings of "A pathological slide" through a FRESH nn.Linear(768, 1280) (:193-197; at the reference's call
    site `unet` is the text-alignment wrapper, segmentor.py:183-205 = ldiffusion_amd.segmentor.TextAlignedUNet, which ignores embeddings
    whose width is not the UNet's and uses its cached ones: that is what happens here for SD-v1.5) -> one UNet pass -> scheduler.step ->
    decode_latents -> numpy_to_pil -> save as PNG (:199-206).  Otherwise a plain copy (:207-208)."""
    if not use_diffusion:
        shutil.copy(src_path, dst_path)
        return
    from PIL import Image
    device = pipeline.vae.device if hasattr(pipeline.vae, "device") else torch.device("cuda", torch.cuda.current_device())
    x = torch.from_numpy(np.asarray(img.convert("RGB").resize((1024, 1024), Image.BILINEAR), np.float32) / 255.0).permute(2, 0, 1)[None].to(device)
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    linear_layer = nn.Linear(768, 1280).to(device)                                               # :193 (randomly initialised on every call, as there)
    ids = torch.tensor(pipeline.tokenizer(["A pathological slide"] * 1)["input_ids"], dtype=torch.long, device=device)
    emb = pipeline.text_encoder(ids)["last_hidden_state"].to(device=device, dtype=torch.float32)
    emb = linear_layer(emb).clone().detach()
    base = getattr(unet, "base_unet", unet)
    if hasattr(unet, "base_unet"):                                                               # the wrapper's rule (segmentor.py:190-202), applied here
        if emb.shape[-1] != unet.cross_attention_dim:                                            # because the fused sampler takes the context up front
            emb = unet.default_text_embeddings
    elif emb.shape[-1] != base.config.cross_attention_dim:
        raise ValueError(f"copy_or_convert_image: text embeddings of width {emb.shape[-1]} for a UNet with cross_attention_dim "
                         f"{base.config.cross_attention_dim} (the reference passes the text-alignment wrapper here)")
    sampler = _sampler(pipeline, base)
    out = sampler.sample((x - mean) / std, emb.to(device=device, dtype=torch.float32)[:1], 1, want_features=False, want_rgb=True)
    rgb = out["rgb"][0].cpu().numpy()
    sampler.check_finite()   # fp16 overflow in either graph raises instead of writing a garbage PNG
    Image.fromarray(rgb).save(dst_path)
