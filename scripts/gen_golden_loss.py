"""Generate tests/golden/reference_infonce.npz by running the REFERENCE's own `InfoNceLoss.compute_contrastive_loss`
(/root/reference/model/loss.py:44-109) in the build container (third-party imports stubbed; the object is built with __new__ so that the
VGG19 download of __init__ is skipped).  Inputs + outputs only are committed; the product's mirror (ldiffusion_amd/loss.py) must
reproduce the loss values with the same torch seed, i.e. draw the same sample triples from the same random stream."""
import importlib.util
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ["diffusers", "torchvision", "torchvision.models"]:
    sys.modules[name] = MagicMock()
spec = importlib.util.spec_from_file_location("ref_loss", "/root/reference/model/loss.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for tag, (B, n, H, W, nlab, seed) in {"a": (1, 5, 64, 64, 4, 11), "b": (2, 3, 64, 64, 7, 12), "c": (1, 2, 40, 40, 2, 13)}.items():
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn((B, n, H, W), generator=g)
    labels = torch.randint(0, nlab, (B, 1, H, W), generator=g).to(torch.uint8)
    if tag == "c":
        labels[:, :, :, :3] = 5            # a small class whose complement exceeds 1024 pixels, and a 2-class rest
    obj = ref.InfoNceLoss.__new__(ref.InfoNceLoss)
    obj.temperature, obj.num_negatives, obj.eps = 0.5, 1024, 1e-8
    torch.manual_seed(100 + seed)
    loss = obj.compute_contrastive_loss(feats.clone().requires_grad_(True), labels)
    out[f"{tag}_features"], out[f"{tag}_labels"], out[f"{tag}_seed"], out[f"{tag}_loss"] = feats.numpy(), labels.numpy(), np.array(100 + seed), np.array(float(loss))
    nxt = torch.rand(1).item()           # the state of the random stream AFTER the call: the mirror must have consumed exactly as much
    out[f"{tag}_next_rand"] = np.array(nxt)
    print(tag, feats.shape, "loss", float(loss), "next rand", nxt)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_infonce.npz"), **out)
