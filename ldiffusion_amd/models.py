"""Drop-in `unet` / `vae` objects backed by the gfx950 HIP library.

They expose exactly the duck-typed surface the reference touches:
  unet(sample, timestep, encoder_hidden_states, ...) -> obj with `.sample` and `[0]`
      /root/reference/segmentor.py:103,205,444,526   ldiffusion.py:160,238   pixel_latent_vector.py:78
  unet.config.cross_attention_dim                     segmentor.py:33,188    ldiffusion.py:142
  UNet2DConditionModel.from_pretrained(dir) / .eval() / .to(device, dtype=) / .save_pretrained(dir)
                                                      segmentor.py:79        ldiffusion.py:139,273
  vae.encode(x).latent_dist.mean / .sample();  vae.decode(z).sample;  .to() / .eval()
                                                      segmentor.py:99,339,379,437,519   ldiffusion.py:228,240
All arithmetic runs in libldiff_hip.so; there is no torch/CPU fallback path.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace

import torch

from . import _lib, configs, weights

_DTYPES = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _load_state_dict(lib, load_fn, handle, sd, expected_names):
    unexpected = [k for k in sd if k not in expected_names]
    if unexpected:
        raise ValueError(f"unexpected tensors in checkpoint: {unexpected[:5]}{' ...' if len(unexpected) > 5 else ''}")
    for name, t in sd.items():
        t = t.detach().to("cpu").contiguous()
        if t.dtype not in _DTYPES:
            t = t.to(torch.float32)
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(load_fn(handle, name.encode(), C.c_void_p(t.data_ptr()), _DTYPES[t.dtype], shape, t.dim()))


class _Output:
    """`.sample` and `[0]`, like diffusers' BaseOutput subclasses."""

    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


class UNet2DConditionModel:
    def __init__(self, cfg: dict, state_dict, device=None):
        _lib.require_gpu()
        cfg = configs.with_defaults(cfg, configs.UNET_DEFAULTS)   # fields a config.json may omit get diffusers' defaults
        configs.validate_unet_config(cfg)
        self._cfg = dict(cfg)
        self.config = SimpleNamespace(**cfg)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = torch.float32
        self._lib = _lib.load()
        c = _lib.UNetCfg()
        c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
        boc = cfg["block_out_channels"]
        c.n_blocks = len(boc)
        for i, v in enumerate(boc):
            c.block_out_channels[i] = v
            c.down_has_attn[i] = int(cfg["down_block_types"][i] == "CrossAttnDownBlock2D")
            c.up_has_attn[i] = int(cfg["up_block_types"][i] == "CrossAttnUpBlock2D")
        c.layers_per_block = cfg["layers_per_block"]
        c.heads = cfg["attention_head_dim"]
        c.cross_attention_dim = cfg["cross_attention_dim"]
        c.norm_num_groups = cfg["norm_num_groups"]
        c.norm_eps = cfg["norm_eps"]
        c.flip_sin_to_cos = int(cfg["flip_sin_to_cos"])
        c.freq_shift = float(cfg["freq_shift"])
        self._h = C.c_void_p()
        _lib.check(self._lib.ldiff_unet_create(C.byref(self._h), C.byref(c), self.device.index or 0))
        self._host_sd = None
        self._ctx_key = None
        self.load_state_dict(state_dict)

    def set_precision(self, mode: int):
        """Storage policy of the graph (include/ldiff.h ldiff_unet_set_precision): 0 all-fp16, 1 split residual stream (default),
        2 every contraction operand split."""
        _lib.check(self._lib.ldiff_unet_set_precision(self._h, int(mode)))
        return self

    def set_graph(self, on: bool):
        """hipGraph replay of the forward's launch sequence (default on; include/ldiff.h ldiff_unet_set_graph)."""
        _lib.check(self._lib.ldiff_unet_set_graph(self._h, int(bool(on))))
        return self

    def check_finite(self):
        """Synchronises the current stream and raises NonFiniteError if a forward enqueued so far produced a non-finite activation
        (fp16 overflow; include/ldiff.h "Non-finite detection")."""
        _lib.check(self._lib.ldiff_unet_check_finite(self._h, _lib.stream_ptr()))
        return self

    @property
    def graph_replays(self) -> int:
        return int(self._lib.ldiff_unet_graph_replays(self._h))

    @property
    def graph_nodes(self) -> int:
        """Kernel launches of the currently captured forward (0 before the first capture)."""
        return int(self._lib.ldiff_unet_graph_nodes(self._h))

    # ---- checkpoint surface ----
    def load_state_dict(self, sd, strict=True):
        _load_state_dict(self._lib, self._lib.ldiff_unet_load, self._h, sd, weights.unet_param_shapes(self._cfg))
        n = self._lib.ldiff_unet_missing(self._h)
        if n and strict:
            names = [self._lib.ldiff_unet_missing_name(self._h, i).decode() for i in range(min(n, 5))]
            raise RuntimeError(f"{n} UNet tensors missing from the checkpoint, e.g. {names}")
        self._host_sd = {k: v.detach().to("cpu") for k, v in sd.items()}
        self._ctx_key = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, device=None, **_ignored):
        if subfolder:
            path = os.path.join(path, subfolder)
        cfg, sd = weights.load_model_dir(path)
        return cls(cfg, sd, device=device)

    def save_pretrained(self, path):
        weights.save_model_dir(path, self._cfg, self._host_sd)

    def state_dict(self):
        return dict(self._host_sd)

    def parameters(self):
        return iter(self._host_sd.values())

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        dt = kwargs.get("dtype", None)
        for a in args:
            if isinstance(a, torch.dtype):
                dt = a
        if dt not in (None, torch.float32):
            raise ValueError("the HIP UNet keeps the reference's float32 boundary (fp16 storage is internal)")
        return self

    # ---- forward ----
    def set_context(self, encoder_hidden_states: torch.Tensor):
        ehs = encoder_hidden_states
        if ehs.dim() != 3 or ehs.shape[-1] != self._cfg["cross_attention_dim"]:
            raise ValueError(f"encoder_hidden_states must be [B, L, {self._cfg['cross_attention_dim']}], got {tuple(ehs.shape)}")
        key = (ehs.data_ptr(), tuple(ehs.shape), ehs._version, ehs.dtype, ehs.device)
        if key == self._ctx_key:
            return
        e = ehs.detach().to(self.device, dtype=torch.float32).contiguous()
        _lib.check(self._lib.ldiff_unet_set_context(self._h, _lib.ptr(e), e.shape[0], e.shape[1], _lib.stream_ptr()))
        self._ctx_key = key
        self._ctx_keepalive = (e, ehs)  # holding the source keeps its storage (and thus the cache key) from being recycled

    def __call__(self, sample, timestep, encoder_hidden_states, *args, **kwargs):
        down_res, mid_res = kwargs.get("down_block_additional_residuals"), kwargs.get("mid_block_additional_residual")
        if sample.dim() != 4 or sample.shape[1] != self._cfg["in_channels"]:
            raise ValueError(f"sample must be [B, {self._cfg['in_channels']}, h, w], got {tuple(sample.shape)}")
        B = sample.shape[0]
        if encoder_hidden_states.shape[0] not in (1, B):
            raise ValueError(f"encoder_hidden_states batch {encoder_hidden_states.shape[0]} does not match sample batch {B}")
        self.set_context(encoder_hidden_states)
        x = sample.detach().to(self.device, dtype=torch.float32).contiguous()
        out = torch.empty((B, self._cfg["out_channels"], x.shape[2], x.shape[3]), device=self.device, dtype=torch.float32)
        if down_res is not None or mid_res is not None:   # ControlNet inputs (segmentor.py:357-375): added to the skips / the mid output
            keep = [t.detach().to(self.device, dtype=torch.float32).contiguous() for t in (down_res or [])]
            shapes = self._skip_shapes(B, x.shape[2], x.shape[3])
            if keep and [tuple(t.shape) for t in keep] != shapes:
                raise ValueError(f"down_block_additional_residuals must have the shapes of the skip tensors {shapes}")
            mid = mid_res.detach().to(self.device, dtype=torch.float32).contiguous() if mid_res is not None else None
            if mid is not None and tuple(mid.shape) != shapes[-1]:
                raise ValueError(f"mid_block_additional_residual must be {shapes[-1]}")
            arr = (C.c_void_p * max(len(keep), 1))(*[t.data_ptr() for t in keep])
            _lib.check(self._lib.ldiff_unet_set_additional_residuals(self._h, arr, len(keep), _lib.ptr(mid)))
            self._residual_keepalive = (keep, mid)
        _lib.check(self._lib.ldiff_unet_forward(self._h, _lib.ptr(x), B, x.shape[2], x.shape[3], float(timestep), _lib.ptr(out), _lib.stream_ptr()))
        return _Output(out)

    forward = __call__

    def _skip_shapes(self, B, h, w):
        """Shapes of the skip tensors in stack order (conv_in, then every resnet/attention output and downsampler of the down path)."""
        boc, lpb = self._cfg["block_out_channels"], self._cfg["layers_per_block"]
        shapes = [(B, boc[0], h, w)]
        for i, c in enumerate(boc):
            shapes += [(B, c, h, w)] * lpb
            if i != len(boc) - 1:
                h, w = h // 2, w // 2
                shapes.append((B, c, h, w))
        return shapes

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ldiff_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _LatentDist:
    """DiagonalGaussianDistribution surface: `.mean`, `.sample()` (segmentor.py:99,339)."""

    def __init__(self, moments):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKL:
    def __init__(self, cfg: dict, state_dict, device=None):
        _lib.require_gpu()
        cfg = configs.with_defaults(cfg, configs.VAE_DEFAULTS)    # e.g. `scaling_factor` (decode_latents reads vae.config.scaling_factor)
        configs.validate_vae_config(cfg)
        self._cfg = dict(cfg)
        self.config = SimpleNamespace(**cfg)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = torch.float32
        self._lib = _lib.load()
        c = _lib.VaeCfg()
        c.in_channels, c.out_channels, c.latent_channels = cfg["in_channels"], cfg["out_channels"], cfg["latent_channels"]
        boc = cfg["block_out_channels"]
        c.n_blocks = len(boc)
        for i, v in enumerate(boc):
            c.block_out_channels[i] = v
        c.layers_per_block = cfg["layers_per_block"]
        c.norm_num_groups = cfg["norm_num_groups"]
        c.scaling_factor = cfg["scaling_factor"]
        self._h = C.c_void_p()
        _lib.check(self._lib.ldiff_vae_create(C.byref(self._h), C.byref(c), self.device.index or 0))
        sd = weights.normalize_vae_keys(state_dict)
        _load_state_dict(self._lib, self._lib.ldiff_vae_load, self._h, sd, weights.vae_param_shapes(self._cfg))
        n = self._lib.ldiff_vae_missing(self._h)
        if n:
            names = [self._lib.ldiff_vae_missing_name(self._h, i).decode() for i in range(min(n, 5))]
            raise RuntimeError(f"{n} VAE tensors missing from the checkpoint, e.g. {names}")
        self._host_sd = {k: v.detach().to("cpu") for k, v in sd.items()}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, device=None, **_ignored):
        if subfolder:
            path = os.path.join(path, subfolder)
        cfg, sd = weights.load_model_dir(path)
        return cls(cfg, sd, device=device)

    def save_pretrained(self, path):
        weights.save_model_dir(path, self._cfg, self._host_sd)

    def set_precision(self, encoder: int = 2, decoder: int = 0):
        """Storage policy of the encoder / decoder graphs (include/ldiff.h ldiff_vae_set_precision).  The current pair is kept in
        `precision` so that a caller that changes it for one call can put back what was there."""
        _lib.check(self._lib.ldiff_vae_set_precision(self._h, int(encoder), int(decoder)))
        self.precision = (int(encoder), int(decoder))
        return self

    def check_finite(self):
        """Synchronises the current stream (and the decode side stream) and raises NonFiniteError if an encode / decode enqueued so far produced
        a non-finite activation (fp16 overflow -- e.g. a decoder fed z / 0.18215 of un-scaled latents; include/ldiff.h "Non-finite detection")."""
        _lib.check(self._lib.ldiff_vae_check_finite(self._h, _lib.stream_ptr()))
        return self

    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    @property
    def scale_factor(self):
        return 2 ** (len(self._cfg["block_out_channels"]) - 1)

    def encode(self, x):
        if x.dim() != 4 or x.shape[1] != self._cfg["in_channels"]:
            raise ValueError(f"image batch must be [B, {self._cfg['in_channels']}, H, W], got {tuple(x.shape)}")
        x = x.detach().to(self.device, dtype=torch.float32).contiguous()
        B, _, H, W = x.shape
        f = self.scale_factor
        mom = torch.empty((B, 2 * self._cfg["latent_channels"], H // f, W // f), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.ldiff_vae_encode(self._h, _lib.ptr(x), B, H, W, _lib.ptr(mom), _lib.stream_ptr()))
        return SimpleNamespace(latent_dist=_LatentDist(mom))

    def _decode(self, z, z_scale, want_sample=False, want_image=False, want_rgb=False, luma=None, slot=0):
        if z.dim() != 4 or z.shape[1] != self._cfg["latent_channels"]:
            raise ValueError(f"latents must be [B, {self._cfg['latent_channels']}, h, w], got {tuple(z.shape)}")
        z = z.detach().to(self.device, dtype=torch.float32).contiguous()
        B, _, h, w = z.shape
        f = self.scale_factor
        H, W = h * f, w * f
        sample = torch.empty((B, self._cfg["out_channels"], H, W), device=self.device, dtype=torch.float32) if want_sample else None
        image = torch.empty((B, H, W, 3), device=self.device, dtype=torch.float32) if want_image else None
        rgb = torch.empty((B, H, W, 3), device=self.device, dtype=torch.uint8) if want_rgb else None
        n_slots = luma.shape[1] if luma is not None else 0
        _lib.check(self._lib.ldiff_vae_decode(self._h, _lib.ptr(z), B, h, w, float(z_scale), _lib.ptr(sample), _lib.ptr(image), _lib.ptr(rgb),
                                              _lib.ptr(luma), n_slots, slot, _lib.stream_ptr()))
        return sample, image, rgb

    def decode(self, z):
        sample, _, _ = self._decode(z, 1.0, want_sample=True)
        return SimpleNamespace(sample=sample)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ldiff_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass
