"""PNDMScheduler shim: host-side PLMS bookkeeping + one HIP linear-combination kernel per step.

Surface used by the reference:
  .set_timesteps(n, device=)   .timesteps (0-dim int64 tensors when iterated)   .scale_model_input(x, t)
  .step(eps, t, x).prev_sample   .alphas_cumprod[t] / len()
  /root/reference/segmentor.py:100-104,438-445,520-527   ldiffusion.py:198,229-234   pixel_latent_vector.py:74-79
Semantics: diffusers 0.34.0 PNDMScheduler with SD-v1.5's scheduler_config.json (skip_prk_steps, leading
spacing, steps_offset=1, set_alpha_to_one=False), see SURVEY.md 8a R6.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib


class _StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample

    def __getitem__(self, i):
        return (self.prev_sample,)[i]


class PNDMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, **_ignored):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule="scaled_linear", skip_prk_steps=True, set_alpha_to_one=False,
                                      steps_offset=steps_offset, prediction_type="epsilon", timestep_spacing="leading")
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, num_inference_steps, device=None):
        n = int(num_inference_steps)
        self.num_inference_steps = n
        step_ratio = self.config.num_train_timesteps // n  # ZeroDivisionError for n == 0, as in the reference
        ts = (np.arange(0, n) * step_ratio).round() + self.config.steps_offset
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(plms)
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample

    def _coeffs(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        sc, ce = C.c_float(), C.c_float()
        _lib.check(_lib.load().ldiff_pndm_coeffs(float(a_t), float(a_prev), C.byref(sc), C.byref(ce)))
        return sc.value, ce.value

    def step(self, model_output, timestep, sample, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        _lib.require_gpu()
        lib = _lib.load()
        timestep = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev_timestep = timestep - ratio
        eps = model_output.detach().to(dtype=torch.float32).contiguous()
        sample = sample.detach().to(dtype=torch.float32).contiguous()
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_timestep = timestep
            timestep = timestep + ratio
        if len(self.ets) == 1 and self.counter == 0:
            terms = [(1.0, self.ets[-1])]
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            terms = [(0.5, eps), (0.5, self.ets[-1])]
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            terms = [(3 / 2, self.ets[-1]), (-1 / 2, self.ets[-2])]
        elif len(self.ets) == 3:
            terms = [(23 / 12, self.ets[-1]), (-16 / 12, self.ets[-2]), (5 / 12, self.ets[-3])]
        else:
            terms = [(55 / 24, self.ets[-1]), (-59 / 24, self.ets[-2]), (37 / 24, self.ets[-3]), (-9 / 24, self.ets[-4])]
        sc, ce = self._coeffs(timestep, prev_timestep)
        ops = [sample] + [t for _, t in terms]
        coef = [sc] + [ce * w for w, _ in terms]
        if not all(o.is_cuda and o.shape == sample.shape for o in ops):
            raise ValueError("scheduler.step operands must be CUDA tensors of one shape")
        out = torch.empty_like(sample)
        cf = (C.c_float * len(coef))(*coef)
        if os.environ.get("LDIFF_DEBUG"):
            print(f"[scheduler.step] t={timestep} prev_t={prev_timestep} n_ets={len(self.ets)} coef:", " ".join(f"{v:.9g}" for v in cf))
        op = (C.c_void_p * len(ops))(*[o.data_ptr() for o in ops])
        _lib.check(lib.ldiff_pndm_step(cf, op, len(ops), _lib.ptr(out), sample.numel(), _lib.stream_ptr()))
        self.counter += 1
        return _StepOutput(out) if return_dict else (out,)
