"""Generate tests/golden/reference_tiling.json from the REFERENCE's own sliding-window helpers, imported here (read-only,
/root/reference/model/nnunetv2/inference/sliding_window_prediction.py) with the one absent third-party import
(acvl_utils) stubbed.  Run once in the build container; only the resulting data (tile origins, Gaussian importance map
samples) is committed -- /root/reference is never read at test time.  BASELINE.json configs[3] (1024^2 ROI -> 512^2
tiles) is the case the product's tiling has to reproduce."""
import importlib.util
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/model/nnunetv2/inference/sliding_window_prediction.py"

for name in ["acvl_utils", "acvl_utils.cropping_and_padding", "acvl_utils.cropping_and_padding.padding"]:
    sys.modules[name] = MagicMock()
spec = importlib.util.spec_from_file_location("ref_sliding_window", SRC)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

cases = [((1024, 1024), (512, 512), 1.0), ((1024, 1024), (512, 512), 0.5), ((110,), (64,), 0.5), ((512, 512), (512, 512), 0.5),
         ((700, 900), (512, 512), 0.5), ((1000, 1024), (512, 512), 1.0), ((513, 2048), (512, 512), 0.5), ((1536, 640), (512, 512), 0.75),
         ((256, 256), (128, 128), 1.0), ((300, 256), (128, 128), 0.5)]
steps = [dict(image_size=list(i), tile_size=list(t), tile_step_size=s, steps=ref.compute_steps_for_sliding_window(i, t, s)) for i, t, s in cases]
g = ref.compute_gaussian((64, 48), sigma_scale=1.0 / 8, value_scaling_factor=10, dtype=torch.float32, device=torch.device("cpu")).numpy()
out = dict(source=SRC + ":32-56 (compute_steps_for_sliding_window), :10-29 (compute_gaussian)", steps=steps,
           gaussian=dict(tile_size=[64, 48], sigma_scale=0.125, value_scaling_factor=10, values=g.astype(np.float64).round(9).tolist()))
with open(os.path.join(ROOT, "tests", "golden", "reference_tiling.json"), "w") as f:
    json.dump(out, f)
print("wrote reference_tiling.json:", len(steps), "step cases; gaussian", g.shape, float(g.min()), float(g.max()))
