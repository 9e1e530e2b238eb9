"""diagnostic: where one graph-replayed fine-tuning step spends its time (SD-v1.5 width, B=8).  usage: python scripts/train_graph_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import autograd as ag, configs, train, weights
from ldiffusion_amd.scheduler import PNDMScheduler
DEV = "cuda:0"
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
unet, dec = train.TrainableUNet(ucfg, usd, DEV), train.FrozenVAEDecoder(vcfg, vsd, DEV)
del usd, vsd
g = torch.Generator().manual_seed(0)
B = 8
z0 = (torch.randn((B, 4, 8, 8), generator=g) * 0.8).to(DEV)
hidden = (torch.randn((B, 6, 768), generator=g) * 0.5).to(DEV)
proj = ((torch.randn((768, 768), generator=g) / 768 ** 0.5).to(DEV).requires_grad_(True), torch.zeros(768, device=DEV, requires_grad=True))
sch = PNDMScheduler(); sch.set_timesteps(1, device=DEV)
ts = [int(t) for t in sch.timesteps]
eps32 = torch.finfo(torch.float32).eps
u_list = [(torch.rand((B, 4, 8, 8), generator=g) * (2 - eps32) + (eps32 - 1)).to(DEV) for _ in ts]
pairs = [[(int(torch.randint(0, 4096, (1,), generator=g)), int(torch.randint(0, 4096, (1,), generator=g)), torch.randint(0, 4096, (1024,), generator=g).tolist())
          for _ in range(8)] for _ in range(B)]
gs = train.GraphedStep(unet, dec, proj, B, ts, sch.alphas_cumprod, max_triples=256)
state = {}
for _ in range(2):
    train.train_step_graphed(gs, z0, hidden, u_list, pairs, state)
torch.cuda.synchronize()
def timed(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"set_batch {timed(lambda: gs.set_batch(z0, hidden, pairs, u_list)):.2f} ms")
print(f"graph replay (forward + loss + backward + unscale) {timed(lambda: gs.graph.replay()):.2f} ms")
print(f"clip_grad_norm {timed(lambda: train.clip_grad_norm(gs.params, 1.0)):.2f} ms")
print(f"adamw {timed(lambda: ag.adamw_step(gs.params, [p.grad for p in gs.params], state, lr=1e-5)):.2f} ms")
print(f"whole step {timed(lambda: train.train_step_graphed(gs, z0, hidden, u_list, pairs, state)):.2f} ms")
