"""ROI tiling for BASELINE.json configs[3]: a 1024x1024 ROI is cut into 512x512 patches, every patch goes through the
sampler (independent units: they shard over batch entries and over ranks exactly like whole patches), and the per-patch
class logits are merged back into the ROI before the arg-max.

Mirrors the vendored nnU-Net helpers the reference's tissue path runs
(/root/reference/model/nnunetv2/inference/sliding_window_prediction.py:32-56 `compute_steps_for_sliding_window`,
:10-29 `compute_gaussian`; /root/reference/model/nnunetv2/inference/predict_from_raw_data.py:517-524 slicer order,
:566-583 Gaussian-weighted accumulation) with the same names and argument meaning; tensors stay on the device.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
import torch


def compute_steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int], tile_step_size: float) -> List[List[int]]:
    if any(i < t for i, t in zip(image_size, tile_size)):
        raise ValueError("image size must be as large or larger than patch_size")
    if not 0 < tile_step_size <= 1:
        raise ValueError("step_size must be larger than 0 and smaller or equal to 1")
    steps = []
    for I, T in zip(image_size, tile_size):
        n = int(math.ceil((I - T) / (T * tile_step_size))) + 1
        actual = (I - T) / (n - 1) if n > 1 else 0.0
        steps.append([int(np.round(actual * k)) for k in range(n)])
    return steps


def tile_origins(image_hw: Tuple[int, int], tile_hw: Tuple[int, int], tile_step_size: float = 1.0) -> List[Tuple[int, int]]:
    sy, sx = compute_steps_for_sliding_window(image_hw, tile_hw, tile_step_size)
    return [(y, x) for y in sy for x in sx]


def split_tiles(image: torch.Tensor, tile_hw: Tuple[int, int], tile_step_size: float = 1.0):
    """[C, H, W] or [1, C, H, W] -> ([n, C, th, tw] contiguous on the same device, origins)."""
    if image.dim() == 4:
        if image.shape[0] != 1:
            raise ValueError("split_tiles takes one ROI")
        image = image[0]
    if image.dim() != 3:
        raise ValueError("ROI must be [C, H, W]")
    th, tw = tile_hw
    origins = tile_origins(tuple(image.shape[1:]), tile_hw, tile_step_size)
    return torch.stack([image[:, y:y + th, x:x + tw] for y, x in origins], 0).contiguous(), origins


def compute_gaussian(tile_size: Sequence[int], sigma_scale: float = 1.0 / 8, value_scaling_factor: float = 1.0,
                     dtype=torch.float32, device="cpu") -> torch.Tensor:
    """Separable Gaussian of an impulse at the tile centre (truncated at 4 sigma, zero padded), peak = value_scaling_factor,
    zeros lifted to the smallest non-zero value.  Built from the 1-D kernels directly (no scipy at run time)."""
    axes = []
    for n in tile_size:
        sigma = n * sigma_scale
        radius = int(4.0 * sigma + 0.5)
        k = np.exp(-0.5 * (np.arange(-radius, radius + 1) / sigma) ** 2)
        k /= k.sum()
        line = np.zeros(n)
        c = n // 2
        for d in range(-radius, radius + 1):   # correlate the impulse with the (symmetric) kernel
            if 0 <= c + d < n:
                line[c + d] = k[d + radius]
        axes.append(line)
    g = axes[0]
    for a in axes[1:]:
        g = np.multiply.outer(g, a)
    g = g / g.max() * value_scaling_factor
    g[g == 0] = g[g != 0].min()
    return torch.from_numpy(g).to(dtype).to(device)


def merge_tile_logits(tiles: torch.Tensor, origins: Sequence[Tuple[int, int]], image_hw: Tuple[int, int], use_gaussian: bool = True) -> torch.Tensor:
    """[n, C, th, tw] float logits -> [C, H, W]: `logits[sl] += pred * g; n[sl] += g; logits /= n` (fp32 on the tiles' device)."""
    if tiles.dim() != 4 or len(origins) != tiles.shape[0]:
        raise ValueError("tiles must be [n, C, th, tw] with one origin per tile")
    n, C, th, tw = tiles.shape
    g = compute_gaussian((th, tw), 1.0 / 8, 10.0, torch.float32, tiles.device) if use_gaussian else torch.ones((th, tw), device=tiles.device)
    acc = torch.zeros((C,) + tuple(image_hw), dtype=torch.float32, device=tiles.device)
    cnt = torch.zeros(tuple(image_hw), dtype=torch.float32, device=tiles.device)
    for t, (y, x) in zip(tiles, origins):
        acc[:, y:y + th, x:x + tw] += t.float() * g
        cnt[y:y + th, x:x + tw] += g
    if not bool((cnt > 0).all()):
        raise RuntimeError("tiles do not cover the ROI")
    return acc / cnt


def merge_tile_masks(masks: torch.Tensor, origins: Sequence[Tuple[int, int]], image_hw: Tuple[int, int]) -> torch.Tensor:
    """Non-overlapping tiles (tile_step_size 1.0 on a multiple of the tile): [n, th, tw] uint8 -> [H, W] by plain copies."""
    n, th, tw = masks.shape
    out = torch.zeros(tuple(image_hw), dtype=masks.dtype, device=masks.device)
    seen = torch.zeros(tuple(image_hw), dtype=torch.bool, device=masks.device)
    for m, (y, x) in zip(masks, origins):
        if bool(seen[y:y + th, x:x + tw].any()):
            raise ValueError("merge_tile_masks needs non-overlapping tiles; merge logits instead")
        out[y:y + th, x:x + tw] = m
        seen[y:y + th, x:x + tw] = True
    if not bool(seen.all()):
        raise RuntimeError("tiles do not cover the ROI")
    return out
