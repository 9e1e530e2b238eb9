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


def pixel_latent_vector(pipeline, vae=None, unet=None, num_inference_steps=5, train_loader=None, text_embeddings=None, out_dir=None, batch_size=8,
                        sampler=None):
    """Mirror of the reference's entry point B (`pixel_latent_vector(pipeline, vae, unet, num_inference_steps)`, pixel_latent_vector.py:58-102):
    for every (image, label) of `train_loader` (the reference reads a module-level loader of batch size 1) run the N-pass sampler and write
    `eval/vector_set/<yy_mm_dd>/pixel_dict_<i>.csv` -- here `batch_size` images at a time through the device-resident batched sampler
    (`LaplaceSampler`: encode once, N x [UNet, PLMS step, decode, uint8, luma] without touching the host) and the vectorised CSV writer.

    `vae` / `unet` are accepted for signature compatibility (the shim pipeline already holds them).  `text_embeddings`: [1 or B, L, cross_attention_dim];
    default = the prompt "A pathological slide" through `pipeline.tokenizer` / `pipeline.text_encoder` (`:64-67`).  The reference then sends these
    through a freshly initialised `nn.Linear(768, 1280)` per image (`:64,68`), whose output no SD-v1.5 UNet accepts (cross_attention_dim is 768):
    the mirror passes the 768-wide embeddings, which is what the text-alignment wrapper of segmentor.py:183-205 substitutes in the paths that run.
    Returns the list of files written."""
    import os
    from datetime import datetime
    from .pipeline import PROMPT, LaplaceSampler
    if train_loader is None:
        raise RuntimeError("pixel_latent_vector: the reference's dataset pipeline (dataset.py, torchvision) is outside this build; pass train_loader=")
    if num_inference_steps == 2 or num_inference_steps < 1:
        raise ValueError("num_inference_steps must be 1 or >= 3 (set_timesteps(N-1) with N = 2 yields a single pass, pixel_latent_vector.py:74)")
    dev = pipeline.vae.device
    if text_embeddings is None:
        if pipeline.tokenizer is None or pipeline.text_encoder is None:
            raise RuntimeError("pixel_latent_vector: no text_embeddings given and the pipeline has no tokenizer / text_encoder")
        ids = torch.tensor(pipeline.tokenizer([PROMPT])["input_ids"], dtype=torch.long, device=dev)
        with torch.no_grad():
            text_embeddings = pipeline.text_encoder(ids)["last_hidden_state"].to(dtype=torch.float32)
    out_dir = out_dir if out_dir is not None else os.path.join("eval", "vector_set", datetime.now().strftime("%y_%m_%d"))
    os.makedirs(out_dir, exist_ok=True)
    s = sampler if sampler is not None else LaplaceSampler(pipeline)
    written, images, labels, index = [], [], [], 0

    def flush():
        nonlocal index
        if not images:
            return
        x = torch.cat(images, 0).to(dev, dtype=torch.float32)
        out = s.sample(x, text_embeddings.to(dev), num_inference_steps, want_features=True, want_rgb=False)
        s.join()   # a caller's sampler in overlap mode 2 defers the side-stream join: the decodes must have written every plane before the copy
        s.check_finite()   # the features are about to leave the device: an fp16 overflow in either graph raises instead of exporting garbage
        feats = out["features"].cpu()
        for b in range(x.shape[0]):
            lab = labels[b]
            lab = lab[0] if lab.dim() == 3 else lab                     # label[0] of the batch, then its channel 0 (`pixel_values[0][i, j]`, :88,93)
            path = os.path.join(out_dir, f"pixel_dict_{index}.csv")
            write_pixel_csv(path, feats[b], lab)
            written.append(path)
            index += 1
        images.clear(); labels.clear()

    for image, label in train_loader:
        for b in range(image.shape[0]):                                 # any loader batch size; the reference's is 1
            images.append(image[b:b + 1]); labels.append(label[b].detach().cpu())
            if len(images) == batch_size:
                flush()
    flush()
    return written
