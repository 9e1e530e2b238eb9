"""Generate tests/golden/reference_*.{json,npz} by importing the REFERENCE's own python (read-only, /root/reference)
in THIS container.  Run once here; the fixtures (data only) are committed, /root/reference is never read at test time.

The reference cannot be imported as shipped (cv2, tifffile, torchvision, diffusers, deepspeed, nnunetv2 are not
installed).  Third-party modules are therefore replaced by in-memory stubs for the duration of this
script, and /root/reference is aliased as package `LDiffusion` (utils.py:13 imports `LDiffusion.dataset`).
Only reference-OWNED code is executed for the fixtures:
  * utils.micro_dice / mean_iou_and_per_class, evaluate.pixel_accuracy / frequency_weighted_iou   (mask metrics)
  * dataset.pixel_to_label / pixel_to_label_cell / map_mask                                         (label LUTs)
  * Segmentor.ldiffusion_augment (segmentor.py:86-112) driven with the oracle's duck-typed pipeline/unet/vae:
    pins the loop's call order, the timesteps that reach the UNet, the text-embedding plumbing and the
    decode -> PIL -> Resize -> ToTensor tail.  `torchvision.transforms` is stubbed with a minimal PIL-based
    Compose/Resize/ToTensor for this one call (the arithmetic under test is the loop, not torchvision).
"""
import importlib
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def install_stubs():
    for name in ["cv2", "tifffile", "diffusers", "deepspeed", "nnunetv2", "nnunetv2.paths", "matplotlib", "matplotlib.pyplot",
                 "torchvision.models", "cellpose", "cellpose.models"]:
        sys.modules[name] = MagicMock()
    from PIL import Image

    class Resize:
        def __init__(self, size, interpolation=None):
            self.size = size

        def __call__(self, im):
            return im.resize((self.size[1], self.size[0]), Image.BILINEAR)

    class ToTensor:
        def __call__(self, im):
            return torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1).contiguous()

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.Resize, tr.ToTensor, tr.Compose = Resize, ToTensor, Compose
    tr.Normalize = MagicMock()
    tr.InterpolationMode = MagicMock()
    tv.transforms = tr
    tv.models = sys.modules["torchvision.models"]
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr
    # alias /root/reference as package LDiffusion
    spec = importlib.util.spec_from_file_location("LDiffusion", os.path.join(REF, "__init__.py"), submodule_search_locations=[REF])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["LDiffusion"] = pkg
    spec.loader.exec_module(pkg)


def metrics_fixture():
    utils = importlib.import_module("LDiffusion.utils")
    evaluate = importlib.import_module("LDiffusion.evaluate")
    dataset = importlib.import_module("LDiffusion.dataset")
    cases = []
    for seed, (B, Cc, H, W) in enumerate([(1, 7, 32, 32), (2, 6, 17, 23), (1, 11, 8, 8), (1, 3, 4, 4)]):
        g = torch.Generator().manual_seed(seed)
        logits = torch.randn((B, Cc, H, W), generator=g)
        target = torch.randint(0, Cc, (B, H, W), generator=g)
        if seed == 3:
            target[:] = 0  # classes absent from both prediction and target
            logits[:, 0] += 100
        dice_scores, avg = utils.micro_dice(logits, target, num_classes=Cc)
        miou, per = utils.mean_iou_and_per_class(logits, target, Cc)
        pa, _ = evaluate.pixel_accuracy(logits, target, Cc)
        fw = evaluate.frequency_weighted_iou(logits, target, Cc)
        cases.append(dict(seed=seed, shape=[B, Cc, H, W], force_class0=seed == 3, dice_per_class=[float(v) for v in dice_scores],
                          dice=float(avg), miou=float(miou), iou_per_class={str(k): v for k, v in per.items()},
                          pixel_accuracy=float(pa), fw_iou=float(fw)))
    mask = np.array([[0, 60, 120], [180, 255, 7]], dtype=np.uint8)
    lut = dict(pixel_to_label={str(k): v for k, v in dataset.pixel_to_label.items()},
               pixel_to_label_cell={str(k): v for k, v in dataset.pixel_to_label_cell.items()},
               map_mask_in=mask.tolist(), map_mask_out=dataset.map_mask(mask).tolist())
    return dict(metrics=cases, luts=lut)


class FakeTokenizer:
    def __call__(self, prompts, **kw):
        return {"input_ids": [[49406, 320, 24857, 5471, 49407] for _ in prompts]}  # unpadded, L = 2 + n_BPE


class FakeTextEncoder:
    def __init__(self, hidden):
        self.config = types.SimpleNamespace(hidden_size=hidden)
        g = torch.Generator().manual_seed(99)
        self.table = torch.randn((49408, hidden), generator=g) * 0.5

    def __call__(self, ids):
        return {"last_hidden_state": self.table[ids]}

    def to(self, *a, **k):
        return self


def augment_fixture():
    from ldiffusion_amd import configs, weights
    from oracle import pipeline as op
    seg_mod = importlib.import_module("LDiffusion.segmentor")
    ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43)
    unet, vae = op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg)
    pipe = op.OraclePipeline(unet, vae, FakeTokenizer(), FakeTextEncoder(48))
    seg = seg_mod.Segmentor(None, None, "cell", 3)
    seg.device = torch.device("cpu")
    g = torch.Generator().manual_seed(7)
    inputs = torch.rand((2, 3, 64, 64), generator=g)
    torch.manual_seed(1)  # nn.Linear(48, 64) init inside _ensure_ldiffusion_proj
    out = seg.ldiffusion_augment(inputs, pipe, unet, vae)   # REFERENCE CODE (segmentor.py:86-112)
    proj = {k: v.clone() for k, v in seg.ldiffusion_proj.state_dict().items()}
    pooled = torch.nn.functional.avg_pool2d(out, 64)  # [2,3,16,16] summary of the [2,3,1024,1024] result
    return dict(unet_calls=list(unet.calls), out_shape=list(out.shape), pooled=pooled.numpy(), proj_weight=proj["weight"].numpy(),
                proj_bias=proj["bias"].numpy(), inputs_seed=7, ids=[49406, 320, 24857, 5471, 49407], hidden=48)


def main():
    os.makedirs(OUT, exist_ok=True)
    install_stubs()
    fx = metrics_fixture()
    with open(os.path.join(OUT, "reference_metrics.json"), "w") as f:
        json.dump(fx, f, indent=1)
    a = augment_fixture()
    np.savez_compressed(os.path.join(OUT, "reference_augment_v3.npz"), unet_calls=np.array(a["unet_calls"]), out_shape=np.array(a["out_shape"]),
                        pooled=a["pooled"], proj_weight=a["proj_weight"], proj_bias=a["proj_bias"], ids=np.array(a["ids"]),
                        hidden=np.array(a["hidden"]), inputs_seed=np.array(a["inputs_seed"]))
    print("wrote", os.listdir(OUT))
    print(json.dumps(fx["metrics"][0], indent=0)[:400])
    print("unet calls", a["unet_calls"], "out", a["out_shape"])


if __name__ == "__main__":
    main()
