"""20 SD-v1.5 UNet passes (B=8, 64x64 latents, default precision 1, hipGraph replay) and nothing else: run under
`rocprofv3 --kernel-trace --stats` to compare the sum of kernel durations with the wall time per pass (what is left is launch gaps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import configs, weights, _lib
if os.environ.get("LDIFF_LIB"):   # an A/B build of the library (diagnostic)
    _lib.LIB_PATH = os.path.abspath(os.environ["LDIFF_LIB"])
from ldiffusion_amd.models import UNet2DConditionModel
ucfg = configs.SD15_UNET
unet = UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), "cuda:0")
B = int(os.environ.get("LDIFF_UNET_B", "8"))   # batch (default 8; 1 = the reference's own batch)
lat = torch.randn((B, 4, 64, 64), device="cuda:0")
ctx = torch.randn((1, 6, 768), device="cuda:0") * 0.5
if os.environ.get("LDIFF_UNET_EAGER"):   # eager launches instead of hipGraph replay (diagnostic)
    unet.set_graph(False)
for _ in range(3):
    unet(lat, 501, ctx)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for _ in range(N):
    unet(lat, 501, ctx)
torch.cuda.synchronize()
print(f"unet step (B={B}): wall per pass {(time.perf_counter() - t0) / N * 1e3:.2f} ms over {N} passes ({N + 3} passes in the process); {unet.graph_nodes} launches per pass (hipGraph nodes)")
