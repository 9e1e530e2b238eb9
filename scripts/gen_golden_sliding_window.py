"""Generate tests/golden/reference_sliding_window.npz by running the REFERENCE's own sliding-window predictor code
(/root/reference/model/nnunetv2/inference/predict_from_raw_data.py:505-589: slicer order, mirroring TTA
`_internal_maybe_mirror_and_predict`, fp16 Gaussian-weighted accumulation `_internal_predict_sliding_window_return_logits`, with the
real `compute_gaussian` default dtype float16 of sliding_window_prediction.py:10-29) in the build container, with the absent
third-party imports stubbed and a tiny deterministic stand-in network.  Only inputs and outputs are committed; /root/reference is
never read at test time."""
import importlib.util
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/model"


class _Stub(MagicMock):
    pass


for name in ["acvl_utils", "acvl_utils.cropping_and_padding", "acvl_utils.cropping_and_padding.padding", "batchgenerators",
             "batchgenerators.dataloading", "batchgenerators.dataloading.multi_threaded_augmenter", "batchgenerators.utilities",
             "batchgenerators.utilities.file_and_folder_operations", "nnunetv2.configuration", "nnunetv2.inference.data_iterators",
             "nnunetv2.inference.export_prediction", "nnunetv2.utilities.file_path_utilities", "nnunetv2.utilities.find_class_by_name",
             "nnunetv2.utilities.helpers", "nnunetv2.utilities.json_export", "nnunetv2.utilities.label_handling",
             "nnunetv2.utilities.label_handling.label_handling", "nnunetv2.utilities.plans_handling",
             "nnunetv2.utilities.plans_handling.plans_handler", "nnunetv2.utilities.utils", "cv2", "tifffile", "nnunetv2.paths", "torchvision", "torchvision.transforms",
             "torchvision.models", "torchvision.models.segmentation", "torchvision.models.segmentation.deeplabv3"]:
    if name not in sys.modules:
        sys.modules[name] = _Stub()
for pkg, path in [("nnunetv2", REF + "/nnunetv2"), ("nnunetv2.inference", REF + "/nnunetv2/inference"), ("nnunetv2.utilities", REF + "/nnunetv2/utilities")]:
    m = types.ModuleType(pkg)
    m.__path__ = [path]
    sys.modules[pkg] = m


def _load(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


sw = _load("nnunetv2.inference.sliding_window_prediction", REF + "/nnunetv2/inference/sliding_window_prediction.py")
sys.modules["nnunetv2.utilities.helpers"].empty_cache = lambda d: None
sys.modules["nnunetv2.utilities.helpers"].dummy_context = __import__("contextlib").nullcontext
pr = _load("nnunetv2.inference.predict_from_raw_data", REF + "/nnunetv2/inference/predict_from_raw_data.py")
P = pr.nnUNetPredictor

C_IN, C_OUT = 3, 4
g = torch.Generator().manual_seed(0)
W = torch.randint(-3, 4, (C_OUT, C_IN), generator=g).float()        # integer weights: the stand-in head is exact in fp32 on any device


def network(x):            # [1, C_IN, h, w] -> [1, C_OUT, h, w]; NOT mirror-equivariant (the column ramp), so TTA changes the result
    ramp = torch.arange(x.shape[-1], dtype=torch.float32, device=x.device) * 0.125
    return torch.einsum("oc,bchw->bohw", W.to(x.device), x.float()) + ramp[None, None, None, :]


out = {}
for tag, (H, Wd, tile, step, mirror) in {"a": (40, 56, (32, 32), 0.5, (0, 1)), "b": (64, 64, (32, 32), 1.0, None), "c": (33, 70, (32, 32), 0.5, (1,))}.items():
    data = torch.randint(0, 16, (C_IN, H, Wd), generator=g).float()      # integer-valued "image"
    fake = types.SimpleNamespace(
        network=network, device=torch.device("cpu"), verbose=False, allow_tqdm=False, use_gaussian=True, use_mirroring=mirror is not None,
        allowed_mirroring_axes=mirror, tile_step_size=step, perform_everything_on_device=True,
        configuration_manager=types.SimpleNamespace(patch_size=list(tile)), label_manager=types.SimpleNamespace(num_segmentation_heads=C_OUT))
    fake._internal_maybe_mirror_and_predict = lambda x, f=fake: P._internal_maybe_mirror_and_predict(f, x)
    data4 = data[:, None]                                    # nnU-Net's 2-D configurations carry a dummy z axis: [C, 1, H, W]
    slicers = P._internal_get_sliding_window_slicers(fake, data4.shape[1:])
    logits = P._internal_predict_sliding_window_return_logits(fake, data4, slicers, True)[:, 0]
    assert logits.dtype == torch.half
    out[f"{tag}_image"] = data.numpy()
    out[f"{tag}_logits_f16"] = logits.numpy()
    out[f"{tag}_cfg"] = np.array([tile[0], tile[1], step, -1 if mirror is None else sum(1 << m for m in mirror)], np.float64)
    out[f"{tag}_n_slicers"] = np.array(len(slicers))
    print(tag, data.shape, "tiles", len(slicers), "logits", tuple(logits.shape), float(logits.float().abs().max()))
g16 = sw.compute_gaussian((32, 32), sigma_scale=1.0 / 8, value_scaling_factor=10, device=torch.device("cpu"))     # the reference's real call: dtype default fp16
out["gaussian_f16_32x32"] = g16.numpy()
out["head_weight"] = W.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_sliding_window.npz"), **out)
print("wrote reference_sliding_window.npz; gaussian dtype", g16.dtype, "min", float(g16.min()), "zeros lifted to", float(g16[g16 > 0].min()))
