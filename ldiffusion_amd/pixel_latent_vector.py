"""Per-pixel latent-vector export: the tail of the reference's canonical N-pass script
(/root/reference/pixel_latent_vector.py:85-101).  The sampler (`LaplaceSampler.sample`) already returns the N luma planes of
every patch as one uint8 tensor [B, N, H, W]; this module turns one patch into the table / CSV file the reference writes, without
the 262,144-iteration Python dict loop (`:92-96`), byte for byte the same file.
"""
from __future__ import annotations

import numpy as np
import torch


def generate_title(n):
    """pixel_latent_vector.py:48-55."""
    title = ["Pixel No."]
    for i in range(n):
        title.append("Sample " + str(i + 1))
    title.append("Category")
    return title


def pixel_table(features, label) -> np.ndarray:
    """features: uint8 [N, H, W] (one patch of `sample()["features"]`, tensor or array); label: [H, W] integer map.
    Returns int64 [H*W, N+1]: row (i*W + j) = the N grey levels of pixel (i, j) followed by its label (the reference's
    `pixel_dict[(i, j)]`, in its insertion order)."""
    f = features.detach().cpu().numpy() if isinstance(features, torch.Tensor) else np.asarray(features)
    lab = label.detach().cpu().numpy() if isinstance(label, torch.Tensor) else np.asarray(label)
    if f.ndim != 3 or lab.shape != f.shape[1:]:
        raise ValueError(f"features must be [N,H,W] and label [H,W]; got {f.shape} and {lab.shape}")
    n, h, w = f.shape
    out = np.empty((h * w, n + 1), dtype=np.int64)
    out[:, :n] = f.reshape(n, h * w).T
    out[:, n] = lab.reshape(-1)
    return out


def write_pixel_csv(path, features, label) -> None:
    """Writes what `csv.writer` produces for the reference's loop (`:98-101`): header, then one row per pixel
    `"(i, j)",v1,...,vN,label` with \\r\\n line ends (file opened with newline='')."""
    tab = pixel_table(features, label)
    f = features.shape
    w = int(f[2])
    idx = np.arange(tab.shape[0])
    keys = np.char.add(np.char.add(np.char.add('"(', (idx // w).astype(str)), np.char.add(", ", (idx % w).astype(str))), ')"')
    cols = [keys] + [tab[:, k].astype(str) for k in range(tab.shape[1])]
    lines = cols[0]
    for c in cols[1:]:
        lines = np.char.add(np.char.add(lines, ","), c)
    with open(path, "w", newline="") as fh:
        fh.write(",".join(generate_title(tab.shape[1] - 1)) + "\r\n")
        fh.write("\r\n".join(lines.tolist()) + "\r\n")
