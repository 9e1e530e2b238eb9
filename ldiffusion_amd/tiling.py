"""ROI tiling for BASELINE.json configs[3]: a 1024x1024 ROI is cut into 512x512 patches, every patch goes through the
sampler (independent units: they shard over batch entries and over ranks exactly like whole patches), and the per-patch
class logits are merged back into the ROI before the arg-max.

Mirrors the vendored nnU-Net helpers the reference's tissue path runs
(/root/reference/model/nnunetv2/inference/sliding_window_prediction.py:32-56 `compute_steps_for_sliding_window`,
:10-29 `compute_gaussian`; /root/reference/model/nnunetv2/inference/predict_from_raw_data.py:505-524 slicer order,
:530-545 mirroring test-time augmentation, :547-589 Gaussian-weighted accumulation in float16, :591-635 padding of images smaller
than a tile) with the same names and argument meaning; tensors stay on the device.  `predict_sliding_window_return_logits` is
pinned bit for bit to the reference's own code (tests/golden/reference_sliding_window.npz, scripts/gen_golden_sliding_window.py).
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
                     dtype=torch.float16, device="cpu") -> torch.Tensor:
    """Separable Gaussian of an impulse at the tile centre (truncated at 4 sigma, zero padded), peak = value_scaling_factor, cast
    to `dtype` (the reference's default is float16) and THEN zeros lifted to the smallest non-zero value, in that order
    (sliding_window_prediction.py:22-27: the float16 tail underflows to zero before the lift).  Built from the 1-D kernels
    directly (no scipy at run time)."""
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
    g = torch.from_numpy(g / g.max() * value_scaling_factor).to(dtype).to(device)
    g[g == 0] = g[g != 0].min()
    return g


def _window_accumulate(acc: torch.Tensor, cnt: torch.Tensor, pred: torch.Tensor, g, y: int, x: int) -> None:
    """acc[:, y:y+th, x:x+tw] += pred * g;  cnt[y:y+th, x:x+tw] += g  (g None: weight 1).  Device tensors: ONE launch of
    ldiff_window_accumulate (the same two roundings per element as the tensor formulation below, which serves host tensors: the CPU tests
    pin it to the reference's predictor)."""
    th, tw = pred.shape[-2:]
    if tuple(pred.shape) != (acc.shape[0], th, tw):   # the kernel indexes pred with acc's class count: a head with fewer channels would be read out of bounds
        raise ValueError(f"window accumulate: prediction {tuple(pred.shape)} does not match the accumulator's {acc.shape[0]} classes")
    if g is not None and tuple(g.shape) != (th, tw):
        raise ValueError(f"window accumulate: importance map {tuple(g.shape)} does not match the tile {(th, tw)}")
    if y < 0 or x < 0 or y + th > acc.shape[1] or x + tw > acc.shape[2]:
        raise ValueError(f"window accumulate: tile {(th, tw)} at {(y, x)} leaves the {tuple(acc.shape[1:])} accumulator")
    if acc.is_cuda:
        from . import _lib
        if acc.dtype == torch.float32 and pred.dtype in (torch.float16, torch.bfloat16):
            pred = pred.float()   # a half-precision head (autocast) into float32 accumulators: identical to the tensor formulation's type promotion
        kinds = {(torch.float32, torch.float32): 0, (torch.float16, torch.float16): 1, (torch.float16, torch.float32): 3}
        if acc.dtype != cnt.dtype or (acc.dtype, pred.dtype) not in kinds or not (acc.is_contiguous() and cnt.is_contiguous()):
            raise ValueError("window accumulate: contiguous float32 accumulators with a float32 prediction, or float16 ones with a float16 / float32 prediction")
        pred = pred.contiguous()
        gg = None if g is None else g.to(acc.dtype).contiguous()
        _lib.check(_lib.load().ldiff_window_accumulate(_lib.ptr(acc), _lib.ptr(cnt), _lib.ptr(pred), _lib.ptr(gg), acc.shape[0], acc.shape[1], acc.shape[2],
                                                       th, tw, int(y), int(x), kinds[(acc.dtype, pred.dtype)], _lib.stream_ptr()))
        return
    acc[:, y:y + th, x:x + tw] += pred * g if g is not None else pred
    cnt[y:y + th, x:x + tw] += g if g is not None else 1


def merge_tile_logits(tiles: torch.Tensor, origins: Sequence[Tuple[int, int]], image_hw: Tuple[int, int], use_gaussian: bool = True,
                      dtype=torch.float32) -> torch.Tensor:
    """[n, C, th, tw] logits -> [C, H, W]: `logits[sl] += pred * g; n[sl] += g; logits /= n` on the tiles' device.
    dtype=torch.float16 reproduces the reference's accumulators and importance map (predict_from_raw_data.py:563-570); the default
    float32 is a deliberate deviation for the sampler's own tile merge (BASELINE configs[3]): it avoids the fp16 underflow of the
    map's tail and the 11-bit accumulation, and is what the oracle's float64 merge is compared with."""
    if tiles.dim() != 4 or len(origins) != tiles.shape[0]:
        raise ValueError("tiles must be [n, C, th, tw] with one origin per tile")
    n, C, th, tw = tiles.shape
    g = compute_gaussian((th, tw), 1.0 / 8, 10.0, dtype, tiles.device) if use_gaussian else torch.ones((th, tw), dtype=dtype, device=tiles.device)
    acc = torch.zeros((C,) + tuple(image_hw), dtype=dtype, device=tiles.device)
    cnt = torch.zeros(tuple(image_hw), dtype=dtype, device=tiles.device)
    for t, (y, x) in zip(tiles, origins):
        _window_accumulate(acc, cnt, t if (dtype == torch.float16 and t.dtype == torch.float32) else t.to(dtype), g, y, x)
    if not bool((cnt > 0).all()):
        raise RuntimeError("tiles do not cover the ROI")
    return acc / cnt


def maybe_mirror_and_predict(network, x: torch.Tensor, mirror_axes=None) -> torch.Tensor:
    """predict_from_raw_data.py:530-545: the network on x and on every non-empty combination of flips over `mirror_axes`
    (0 = rows, 1 = columns of a [1, C, h, w] tile), each prediction flipped back, averaged."""
    import itertools
    prediction = network(x)
    if mirror_axes is not None:
        if max(mirror_axes) > x.dim() - 3:
            raise ValueError("mirror_axes does not match the dimension of the input!")
        combos = [c for i in range(len(mirror_axes)) for c in itertools.combinations([m + 2 for m in mirror_axes], i + 1)]
        for axes in combos:
            prediction += torch.flip(network(torch.flip(x, (*axes,))), (*axes,))
        prediction /= (len(combos) + 1)
    return prediction


def pad_to_tile(image: torch.Tensor, tile_hw: Tuple[int, int]):
    """acvl_utils `pad_nd_image(image, new_shape=tile, 'constant', value 0, return_slicer=True)` for [C, H, W]: images smaller than
    the tile are zero padded, the excess split evenly (the odd element goes to the far side).  Returns (padded, (sy, sx))."""
    C, H, W = image.shape
    ph, pw = max(tile_hw[0] - H, 0), max(tile_hw[1] - W, 0)
    t, l = ph // 2, pw // 2
    if ph or pw:
        image = torch.nn.functional.pad(image, (l, pw - l, t, ph - t), mode="constant", value=0)
    return image, (slice(t, t + H), slice(l, l + W))


@torch.no_grad()
def predict_sliding_window_return_logits(image: torch.Tensor, network, num_heads: int, tile_hw: Tuple[int, int], tile_step_size: float = 0.5,
                                         use_gaussian: bool = True, mirror_axes=None, acc_dtype=torch.float16) -> torch.Tensor:
    """nnUNetPredictor.predict_sliding_window_return_logits (predict_from_raw_data.py:547-635) for one 2-D image [C, H, W] that
    is already on the device: tiles in the reference's slicer order (rows outer), mirroring TTA per tile, Gaussian-weighted
    accumulation `logits[sl] += pred * g; n[sl] += g` in `acc_dtype` (float16 in the reference), `logits /= n`, the inf check,
    padding reverted.  `network([1, C, th, tw]) -> [1, num_heads, th, tw]` is the injected tissue head.  Returns
    [num_heads, H, W] in `acc_dtype` on the image's device."""
    if image.dim() != 3:
        raise ValueError("input_image must be [C, H, W]")
    data, revert = pad_to_tile(image, tile_hw)
    th, tw = tile_hw
    logits = torch.zeros((num_heads,) + tuple(data.shape[1:]), dtype=acc_dtype, device=data.device)
    n_pred = torch.zeros(tuple(data.shape[1:]), dtype=acc_dtype, device=data.device)
    g = compute_gaussian((th, tw), sigma_scale=1.0 / 8, value_scaling_factor=10, dtype=acc_dtype, device=data.device) if use_gaussian else None
    for y, x in tile_origins(tuple(data.shape[1:]), tile_hw, tile_step_size):
        pred = maybe_mirror_and_predict(network, data[None, :, y:y + th, x:x + tw], mirror_axes)[0]
        _window_accumulate(logits, n_pred, pred, g if use_gaussian else None, y, x)
    logits /= n_pred
    if bool(torch.isinf(logits).any()):
        raise RuntimeError("Encountered inf in predicted array. Aborting... If this problem persists, reduce value_scaling_factor in "
                           "compute_gaussian or increase the dtype of predicted_logits to fp32")
    return logits[(slice(None),) + revert]


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
