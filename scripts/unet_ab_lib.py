import os, sys, time
sys.path.insert(0, os.getcwd())
from ldiffusion_amd import _lib
if os.environ.get("LDIFF_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["LDIFF_LIB"])
import torch
from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import UNet2DConditionModel
ucfg = configs.SD15_UNET
unet = UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), "cuda:0")
ctx = torch.randn((1, 6, 768), device="cuda:0") * 0.5
for B in (8, 1):
    lat = torch.randn((B, 4, 64, 64), device="cuda:0")
    for _ in range(3): unet(lat, 501, ctx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): unet(lat, 501, ctx)
    torch.cuda.synchronize(); print(os.environ.get("LDIFF_LIB", "product"), "B", B, "unet pass ms", round((time.perf_counter() - t0) / 20 * 1e3, 3))
