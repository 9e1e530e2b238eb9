"""Oracle: Laplace noise, decode post-processing, luma features, mask tail.
TEST INFRASTRUCTURE ONLY.

Every function cites the reference (or third-party) line it follows.  R-numbers
refer to SURVEY.md section 8a.
"""
from __future__ import annotations

import numpy as np
import torch

SCALING_FACTOR = 0.18215  # vae/config.json scaling_factor (SD-v1.5)


# ---- R7: torch.distributions.Laplace.rsample (pinned vs installed torch) -------------------
def laplace_from_uniform(u: torch.Tensor, loc, scale) -> torch.Tensor:
    """x = loc - scale * sign(u) * log1p(-|u|),  u ~ U(eps-1, 1).
    torch/distributions/laplace.py rsample(); reference use: ldiffusion.py:234-237
    (`Laplace(0, sqrt(1-abar_t)).sample(latents.shape)`), segmentor.py:344-345."""
    return loc - scale * u.sign() * torch.log1p(-u.abs())


def laplace_uniform_draw(shape, generator: torch.Generator) -> torch.Tensor:
    """The `u` draw of Laplace.rsample: uniform_(finfo.eps - 1, 1) in fp32."""
    finfo = torch.finfo(torch.float32)
    u = torch.empty(shape, dtype=torch.float32)
    return u.uniform_(finfo.eps - 1, 1, generator=generator)


def laplace_forward_noise(z0: torch.Tensor, abar_t: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """ldiffusion.py:234-237: noisy = z0 + Laplace(0, sqrt(1 - abar_t)).sample()  (given u)."""
    scale = torch.sqrt(1 - abar_t).to(torch.float32)
    return (z0 + laplace_from_uniform(u, 0.0, scale)).to(torch.float32)


# ---- R8: decode_latents tail + numpy_to_pil -------------------------------------------------
def decode_post(x: torch.Tensor) -> np.ndarray:
    """StableDiffusionImg2ImgPipeline.decode_latents tail: (x/2+0.5).clamp(0,1) -> NHWC fp32 numpy.
    (reference call sites segmentor.py:106,447,529; pixel_latent_vector.py:81; utils.py:204)"""
    img = (x / 2 + 0.5).clamp(0, 1)
    return img.cpu().permute(0, 2, 3, 1).float().numpy()


def to_uint8(images: np.ndarray) -> np.ndarray:
    """numpy_to_pil: (images*255).round().astype('uint8')  -- numpy round-half-even.
    (segmentor.py:107,448,530; pixel_latent_vector.py:82)"""
    return (images * 255).round().astype("uint8")


# ---- R9: PIL convert("L") (pinned vs installed Pillow) --------------------------------------
def luma_u8(rgb: np.ndarray) -> np.ndarray:
    """ITU-R 601-2 integer luma exactly as Pillow's ImagingConvert rgb2l:
    L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16.   pixel_latent_vector.py:85."""
    r = rgb[..., 0].astype(np.uint32)
    g = rgb[..., 1].astype(np.uint32)
    b = rgb[..., 2].astype(np.uint32)
    return ((19595 * r + 38470 * g + 7471 * b + 0x8000) >> 16).astype(np.uint8)


def luma_float(rgb: torch.Tensor) -> torch.Tensor:
    """ldiffusion.py:241-242: (rgb * [0.2989, 0.5870, 0.1140]).sum(dim=1, keepdim=True)."""
    w = torch.tensor([0.2989, 0.5870, 0.1140], dtype=torch.float32).view(1, 3, 1, 1)
    return (rgb * w).sum(dim=1, keepdim=True)


# ---- F13: mask tail -------------------------------------------------------------------------
def argmax_mask(logits: torch.Tensor) -> np.ndarray:
    """segmentor.py:536-537: argmax(softmax(out,1),1) -> uint8.  Softmax is monotone so the
    argmax of the logits is identical; ties resolve to the lowest class index (torch.argmax)."""
    return torch.argmax(torch.softmax(logits, dim=1), dim=1).cpu().numpy().astype(np.uint8)


def probe_argmax(features_u8: np.ndarray, weight: np.ndarray, bias, scale: float = 1.0 / 255.0) -> np.ndarray:
    """Linear probe over the per-pixel latent vectors (pixel_latent_vector.py:85-93: the N luma planes of a pixel) followed by the mask
    tail of segmentor.py:536-537, in the arithmetic `ldiff_probe_argmax_u8` pins (include/ldiff.h): float32 throughout,
    x_n = f_n * scale; acc = bias_c; acc = acc + w[c,n] * x_n in plane order, every operation rounded (no fused multiply-add);
    first maximal class.  features [B,N,H,W] uint8, weight [C,N], bias [C] or None -> uint8 [B,H,W]."""
    f = features_u8.astype(np.float32) * np.float32(scale)
    w = np.asarray(weight, dtype=np.float32)
    C, N = w.shape
    best, idx = None, None
    for c in range(C):
        acc = np.full(f[:, 0].shape, np.float32(0.0 if bias is None else bias[c]), dtype=np.float32)
        for n in range(N):
            acc = (acc + (w[c, n] * f[:, n]).astype(np.float32)).astype(np.float32)
        if c == 0:
            best, idx = acc.copy(), np.zeros(acc.shape, dtype=np.uint8)
        else:
            m = acc > best
            best = np.where(m, acc, best)
            idx = np.where(m, np.uint8(c), idx)
    return idx
