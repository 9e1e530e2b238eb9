"""TEST INFRASTRUCTURE (CPU oracle, see oracle/__init__.py): restatement of the reference's ROI tiling.

BASELINE.json configs[3] tiles a 1024x1024 ROI into 512x512 patches.  The reference does this with the vendored
nnU-Net helpers (/root/reference/model/nnunetv2/inference/sliding_window_prediction.py:32-56 tile origins, :10-29 Gaussian
importance map; /root/reference/model/nnunetv2/inference/predict_from_raw_data.py:505-524 row-major slicer order,
:566-583 weighted accumulation `logits[sl] += pred * g; n[sl] += g; logits /= n`).
PINNED: tests/golden/reference_tiling.json holds the outputs of those reference functions run in the build container
(scripts/gen_golden_tiling.py); tests/test_cpu_oracle.py checks this file against it.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np


def steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int], tile_step_size: float) -> List[List[int]]:
    """Tile origins per axis (sliding_window_prediction.py:32-56): ceil((I-T)/(T*s))+1 tiles, evenly spread over [0, I-T],
    rounded half-to-even."""
    if any(i < t for i, t in zip(image_size, tile_size)):
        raise ValueError("image size must be as large or larger than the tile size")
    if not 0 < tile_step_size <= 1:
        raise ValueError("step_size must be larger than 0 and smaller or equal to 1")
    out = []
    for I, T in zip(image_size, tile_size):
        n = int(math.ceil((I - T) / (T * tile_step_size))) + 1
        span = I - T
        step = span / (n - 1) if n > 1 else 0.0
        out.append([int(np.round(step * k)) for k in range(n)])
    return out


def tile_origins(image_hw: Tuple[int, int], tile_hw: Tuple[int, int], tile_step_size: float) -> List[Tuple[int, int]]:
    """Row-major (y outer, x inner) origins, the slicer order of predict_from_raw_data.py:517-524."""
    sy, sx = steps_for_sliding_window(image_hw, tile_hw, tile_step_size)
    return [(y, x) for y in sy for x in sx]


def gaussian_importance(tile_size: Sequence[int], sigma_scale: float = 1.0 / 8, value_scaling_factor: float = 1.0) -> np.ndarray:
    """sliding_window_prediction.py:10-29: unit impulse at the centre, separable Gaussian filter (scipy semantics:
    truncate 4 sigma, zero padding), normalised to `value_scaling_factor` at the maximum, zeros lifted to the smallest
    non-zero value."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(tuple(tile_size))
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode="constant", cval=0)
    g = g / g.max() * value_scaling_factor
    g[g == 0] = g[g != 0].min()
    return g


def merge_logits(tiles: np.ndarray, origins: Sequence[Tuple[int, int]], image_hw: Tuple[int, int], use_gaussian: bool = True) -> np.ndarray:
    """predict_from_raw_data.py:566-583 for 2-D tiles [n, C, th, tw] -> [C, H, W] (float64 accumulation here)."""
    n, C, th, tw = tiles.shape
    g = gaussian_importance((th, tw), 1.0 / 8, 10.0) if use_gaussian else np.ones((th, tw))
    acc = np.zeros((C,) + tuple(image_hw))
    cnt = np.zeros(tuple(image_hw))
    for t, (y, x) in zip(tiles, origins):
        acc[:, y:y + th, x:x + tw] += t.astype(np.float64) * g
        cnt[y:y + th, x:x + tw] += g
    return acc / cnt
