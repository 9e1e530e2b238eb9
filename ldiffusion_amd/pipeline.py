"""`pipeline` shim (StableDiffusionImg2ImgPipeline surface) and the batched Laplace-diffusion sampler.

Surface used by the reference:
  pipeline.vae / .unet / .scheduler / .tokenizer / .text_encoder / .decode_latents(z) / .numpy_to_pil(np) / .to(device)
  /root/reference/segmentor.py:32,57-59,77-80,106-107   ldiffusion.py:67-69,141,213-216   pixel_latent_vector.py:41-47,81-82
`LaplaceSampler` is the batched, device-resident form of the per-image loop bodies
  pixel_latent_vector.py:72-93 (N passes, luma feature per pass) and segmentor.py:99-107 / 519-530 (one pass),
running entirely inside libldiff_hip.so (`ldiff_sample`).
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib, weights
from .models import AutoencoderKL, UNet2DConditionModel
from .scheduler import PNDMScheduler

PROMPT = "A pathological slide"  # segmentor.py:92, ldiffusion.py:210, pixel_latent_vector.py:65


class StableDiffusionImg2ImgPipeline:
    def __init__(self, vae: AutoencoderKL, unet: UNet2DConditionModel, scheduler=None, tokenizer=None, text_encoder=None):
        self.vae, self.unet = vae, unet
        self.scheduler = scheduler if scheduler is not None else PNDMScheduler()
        self.tokenizer, self.text_encoder = tokenizer, text_encoder
        self.device = vae.device

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, device=None, **_ignored):
        """Reads the diffusers SD directory layout: {unet,vae}/(config.json + safetensors), scheduler/scheduler_config.json,
        and tokenizer/ + text_encoder/ through `transformers` when those folders exist (CLIP is outside the kernel scope)."""
        if torch_dtype not in (None, torch.float32):
            raise ValueError("the reference loads the pipeline in float32 (ldiffusion.py:67); other dtypes are not supported")
        unet = UNet2DConditionModel.from_pretrained(os.path.join(path, "unet"), device=device)
        vae = AutoencoderKL.from_pretrained(os.path.join(path, "vae"), device=device)
        sched_cfg = {}
        sp = os.path.join(path, "scheduler", "scheduler_config.json")
        if os.path.exists(sp):
            with open(sp) as f:
                sched_cfg = {k: v for k, v in json.load(f).items() if k in ("num_train_timesteps", "beta_start", "beta_end", "steps_offset")}
        tok = enc = None
        if os.path.isdir(os.path.join(path, "tokenizer")) and os.path.isdir(os.path.join(path, "text_encoder")):
            from transformers import CLIPTextModel, CLIPTokenizer
            tok = CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
            enc = CLIPTextModel.from_pretrained(os.path.join(path, "text_encoder")).to(vae.device)
        return cls(vae, unet, PNDMScheduler(**sched_cfg), tok, enc)

    def to(self, *a, **k):
        return self

    def decode_latents(self, latents):
        """(1/scaling_factor * z) -> vae.decode -> (x/2+0.5).clamp(0,1) -> NHWC float32 numpy on the host."""
        _, image, _ = self.vae._decode(latents, 1.0 / self.vae.config.scaling_factor, want_image=True)
        host = image.cpu().numpy()
        self.vae.check_finite()   # the copy above has synchronised already: an fp16 overflow in the decoder raises here instead of returning a garbage image
        return host

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None, ...]
        images = (images * 255).round().astype("uint8")
        return [Image.fromarray(im) for im in images]


class LaplaceSampler:
    """Batched sampler over [B,3,H,W] patches; everything stays on the GPU until the caller reads the outputs."""

    def __init__(self, pipeline: StableDiffusionImg2ImgPipeline):
        _lib.require_gpu()
        self.pipeline = pipeline
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.ldiff_pipeline_create(C.byref(self._h), pipeline.unet._h, pipeline.vae._h))
        abar = pipeline.scheduler.alphas_cumprod.detach().to("cpu", torch.float32).contiguous()
        _lib.check(self._lib.ldiff_pipeline_set_alphas_cumprod(self._h, C.cast(abar.data_ptr(), C.POINTER(C.c_float)), abar.numel()))
        self._pending = None

    def set_overlap(self, mode):
        """0 / False: everything on the current stream; 1 / True (default): VAE decode of pass k on a side stream beside the UNet
        pass k+1, joined before sample() returns; 2: the same with the join deferred to join() -- features / rgb of a sample()
        call must not be read before it (lets the next batch start under the trailing decodes).  Identical results in all modes."""
        _lib.check(self._lib.ldiff_pipeline_set_overlap(self._h, int(mode)))

    def join(self):
        """Make the current stream wait for the decodes of this sampler's last sample() call (mode 2)."""
        _lib.check(self._lib.ldiff_pipeline_join(self._h, _lib.stream_ptr()))
        self._pending = None

    def check_finite(self):
        """Joins a deferred decode, synchronises the current stream and raises NonFiniteError if the UNet or the VAE produced a non-finite
        activation in any sample() so far (include/ldiff.h "Non-finite detection").  Call it where the results leave the device; sample() itself
        never synchronises (a flagged call is otherwise reported by the next call on the handles)."""
        _lib.check(self._lib.ldiff_pipeline_check_finite(self._h, _lib.stream_ptr()))
        self._pending = None

    def timesteps(self, num_inference_steps: int):
        buf = (C.c_int64 * 1024)()
        n = self._lib.ldiff_plms_timesteps(int(num_inference_steps), buf, 1024)
        _lib.check(n if n < 0 else 0)
        return [int(buf[i]) for i in range(n)]

    def sample(self, images: torch.Tensor, encoder_hidden_states: torch.Tensor, num_inference_steps: int,
               want_features=True, want_rgb=True):
        """Returns dict(latents [B,4,h,w] f32, features [B,N,H,W] u8, rgb [B,H,W,3] u8) as CUDA tensors."""
        vae, unet = self.pipeline.vae, self.pipeline.unet
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"images must be [B,3,H,W], got {tuple(images.shape)}")
        x = images.detach().to(vae.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        if encoder_hidden_states.shape[0] not in (1, B):
            raise ValueError("encoder_hidden_states batch must be 1 or the image batch")
        unet.set_context(encoder_hidden_states)
        N = len(self.timesteps(num_inference_steps))
        f = vae.scale_factor
        lat = torch.empty((B, vae.config.latent_channels, H // f, W // f), device=vae.device, dtype=torch.float32)
        feats = torch.empty((B, N, H, W), device=vae.device, dtype=torch.uint8) if want_features else None
        rgb = torch.empty((B, H, W, 3), device=vae.device, dtype=torch.uint8) if want_rgb else None
        _lib.check(self._lib.ldiff_sample(self._h, _lib.ptr(x), B, H, W, int(num_inference_steps), _lib.ptr(lat), _lib.ptr(feats), _lib.ptr(rgb),
                                          _lib.stream_ptr()))
        # deferred join (mode 2): the side stream may still be writing features / rgb after this returns.  Hold a reference until
        # join() (or the next sample(), which joins inside the library) so that a caller who drops the dict early -- e.g. on an
        # exception path -- cannot get the block recycled by the caching allocator under the pending decodes.
        self._pending = (x, feats, rgb)
        return dict(latents=lat, features=feats, rgb=rgb)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ldiff_pipeline_destroy(self._h)   # synchronises the device: pending side-stream decodes finish first
                self._h = None
            self._pending = None
        except Exception:
            pass


def argmax_mask(logits: torch.Tensor) -> torch.Tensor:
    """segmentor.py:536-537 `argmax(softmax(out,1),1)` -> uint8 [B,H,W] on the GPU."""
    _lib.require_gpu()
    if logits.dim() != 4:
        raise ValueError(f"logits must be [B,C,H,W], got {tuple(logits.shape)}")
    x = logits.detach().to(dtype=torch.float32).contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W), device=x.device, dtype=torch.uint8)
    _lib.check(_lib.load().ldiff_argmax_u8(_lib.ptr(x), B, Cc, H, W, _lib.ptr(out), _lib.stream_ptr()))
    return out


def probe_argmax_mask(features: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None, scale: float = 1.0 / 255.0) -> torch.Tensor:
    """Linear probe over the per-pixel latent vectors + the mask tail in one launch (`ldiff_probe_argmax_u8`): features uint8 [B,N,H,W]
    (pixel_latent_vector.py:85-93), weight [C,N], bias [C] -> uint8 [B,H,W] = argmax_c(bias_c + sum_n weight[c,n] * features_n * scale)
    (segmentor.py:536-537), without materialising the float logits."""
    _lib.require_gpu()
    if features.dim() != 4 or features.dtype != torch.uint8:
        raise ValueError(f"features must be uint8 [B,N,H,W], got {features.dtype} {tuple(features.shape)}")
    B, N, H, W = features.shape
    if weight.dim() != 2 or weight.shape[1] != N:
        raise ValueError(f"weight must be [C,{N}], got {tuple(weight.shape)}")
    Cc = weight.shape[0]
    if bias is not None and tuple(bias.shape) != (Cc,):
        raise ValueError(f"bias must be [{Cc}], got {tuple(bias.shape)}")
    f = features.detach().contiguous()
    w = weight.detach().to(f.device, dtype=torch.float32).contiguous()
    bvec = None if bias is None else bias.detach().to(f.device, dtype=torch.float32).contiguous()
    out = torch.empty((B, H, W), device=f.device, dtype=torch.uint8)
    _lib.check(_lib.load().ldiff_probe_argmax_u8(_lib.ptr(f), B, N, H, W, _lib.ptr(w), _lib.ptr(bvec), float(scale), Cc, _lib.ptr(out), _lib.stream_ptr()))
    return out


def laplace_noise(z0: torch.Tensor, scale: float, u: torch.Tensor = None, seed: int = 0, offset: int = 0) -> torch.Tensor:
    """ldiffusion.py:234-237: z0 + Laplace(0, scale).sample() -- given `u` (parity) or from the device Philox stream."""
    _lib.require_gpu()
    z = z0.detach().to(dtype=torch.float32).contiguous()
    out = torch.empty_like(z)
    if u is not None:
        if u.shape != z.shape:
            raise ValueError("u must have the shape of z0")
        u = u.detach().to(z.device, dtype=torch.float32).contiguous()
    _lib.check(_lib.load().ldiff_laplace_add(_lib.ptr(z), float(scale), _lib.ptr(u), int(seed), int(offset), _lib.ptr(out), z.numel(), _lib.stream_ptr()))
    return out


def luma_float(rgb: torch.Tensor) -> torch.Tensor:
    """ldiffusion.py:241-242 float luma, [B,3,H,W] -> [B,1,H,W]."""
    _lib.require_gpu()
    x = rgb.detach().to(dtype=torch.float32).contiguous()
    B, _, H, W = x.shape
    out = torch.empty((B, 1, H, W), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().ldiff_luma_float(_lib.ptr(x), _lib.ptr(out), B, H, W, _lib.stream_ptr()))
    return out


def bilinear_resize(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) for fp32 [B,C,H,W] (ldiffusion.py:240,250)."""
    _lib.require_gpu()
    if x.dim() != 4:
        raise ValueError("bilinear_resize takes [B,C,H,W]")
    x = x.detach().to(dtype=torch.float32).contiguous()
    B, Cc, H, W = x.shape
    oh, ow = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
    out = torch.empty((B, Cc, oh, ow), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().ldiff_bilinear_resize(_lib.ptr(x), _lib.ptr(out), B, Cc, H, W, oh, ow, _lib.stream_ptr()))
    return out


def laplace_features(pipeline: StableDiffusionImg2ImgPipeline, images: torch.Tensor, text_embeddings: torch.Tensor, num_inference_steps: int,
                     u_list=None, seed: int = 0, out_hw: int = 64, decoder_precision: int = 1):
    """The forward part of the reference's training step (ldiffusion.py:228-247): z0 = encode(x).mean is kept
    fixed; per scheduler timestep x_t = z0 + Laplace(0, sqrt(1 - abar_t)), eps = unet(x_t, t, ctx), the UNet output is decoded
    directly, resized to out_hw x out_hw (bilinear) and reduced to a float luma plane; the planes are concatenated.
    `u_list[i]` (optional) is the uniform draw of step i (parity is defined given u); otherwise the device Philox stream.
    Here the decoder's FLOAT output is the feature (no uint8 quantisation behind it as in the sampler), so the decodes run with the
    split residual stream (`decoder_precision`, ldiff_vae_set_precision) instead of the sampler's all-fp16 decoder default."""
    vae, unet, sch = pipeline.vae, pipeline.unet, pipeline.scheduler
    saved = getattr(vae, "precision", (2, 0))   # the caller's setting (the library's defaults if it was never changed)
    vae.set_precision(saved[0], decoder_precision)
    try:
        return _laplace_features(pipeline, images, text_embeddings, num_inference_steps, u_list, seed, out_hw)
    finally:
        vae.set_precision(*saved)


def _laplace_features(pipeline, images, text_embeddings, num_inference_steps, u_list, seed, out_hw):
    vae, unet, sch = pipeline.vae, pipeline.unet, pipeline.scheduler
    z0 = vae.encode(images).latent_dist.mean.to(dtype=torch.float32)
    sch.set_timesteps(num_inference_steps, device=z0.device)
    grays, rgb = [], None
    for i, t in enumerate(sch.timesteps):
        lat = sch.scale_model_input(z0, t)
        scale = float(torch.sqrt(1 - sch.alphas_cumprod[int(t)]))
        noisy = laplace_noise(lat, scale, u=None if u_list is None else u_list[i], seed=seed, offset=i * lat.numel())
        den = unet(noisy, t, text_embeddings).sample
        rgb = bilinear_resize(vae.decode(den).sample, out_hw)
        grays.append(luma_float(rgb))
    return dict(gray=torch.cat(grays, dim=1), rgb=rgb)
