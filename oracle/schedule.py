"""Oracle: PNDM (PLMS) scheduler, restated.  TEST INFRASTRUCTURE ONLY.

Mirrors diffusers 0.34.0 `PNDMScheduler` with the SD-v1.5 scheduler_config.json
(beta_start 0.00085, beta_end 0.012, scaled_linear, 1000 train steps,
skip_prk_steps=True, set_alpha_to_one=False, steps_offset=1, leading spacing,
epsilon prediction).  PARITY UNPINNED (diffusers absent, see oracle/__init__).

Reference call sites this serves (all in /root/reference):
  scheduler.set_timesteps(n)        segmentor.py:100,438,520  pixel_latent_vector.py:74  ldiffusion.py:229
  scheduler.timesteps               segmentor.py:101,442,524  pixel_latent_vector.py:76
  scheduler.scale_model_input       segmentor.py:102,443,525  pixel_latent_vector.py:77  ldiffusion.py:233
  scheduler.step(..).prev_sample    segmentor.py:104,445,527  pixel_latent_vector.py:79
  scheduler.alphas_cumprod[t]       ldiffusion.py:198,234
"""
from __future__ import annotations

import numpy as np
import torch

NUM_TRAIN_TIMESTEPS = 1000
BETA_START = 0.00085
BETA_END = 0.012
STEPS_OFFSET = 1


def alphas_cumprod() -> torch.Tensor:
    """`scaled_linear` betas -> cumulative product of (1-beta), all in fp32 torch
    (PNDMScheduler.__init__)."""
    betas = torch.linspace(BETA_START ** 0.5, BETA_END ** 0.5, NUM_TRAIN_TIMESTEPS, dtype=torch.float32) ** 2
    alphas = 1.0 - betas
    return torch.cumprod(alphas, dim=0)


def plms_timesteps(num_inference_steps: int) -> np.ndarray:
    """PNDMScheduler.set_timesteps with skip_prk_steps=True, 'leading' spacing."""
    n = int(num_inference_steps)
    step_ratio = NUM_TRAIN_TIMESTEPS // n
    ts = (np.arange(0, n) * step_ratio).round()
    ts = ts + STEPS_OFFSET
    plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
    return plms.astype(np.int64)


class _StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample

    def __getitem__(self, i):
        return (self.prev_sample,)[i]


class PNDMOracle:
    """Stateful restatement of PNDMScheduler (PLMS branch only, as SD-v1.5 configures it)."""

    def __init__(self):
        self.alphas_cumprod = alphas_cumprod()
        self.final_alpha_cumprod = self.alphas_cumprod[0]  # set_alpha_to_one=False
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = int(num_inference_steps)
        self.timesteps = torch.from_numpy(plms_timesteps(num_inference_steps))
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def scale_model_input(self, sample, *args, **kwargs):
        return sample  # identity for PNDM

    def step(self, model_output, timestep, sample):
        """step_plms (skip_prk_steps=True => every step is a PLMS step)."""
        timestep = int(timestep)
        ratio = NUM_TRAIN_TIMESTEPS // self.num_inference_steps
        prev_timestep = timestep - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + ratio

        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])

        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return _StepOutput(prev_sample)

    def prev_sample_coeffs(self, timestep: int, prev_timestep: int):
        """(sample_coeff, eps_coeff) of _get_prev_sample as fp32 0-dim tensors:
        prev = sample_coeff * sample - eps_coeff * model_output."""
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return sample_coeff, (a_prev - a_t) / denom, denom, (a_prev - a_t)

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        sample_coeff, _, denom, da = self.prev_sample_coeffs(timestep, prev_timestep)
        return sample_coeff * sample - da * model_output / denom
