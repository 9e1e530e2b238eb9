"""Time one fine-tuning step (ldiffusion_amd/train.py; /root/reference/ldiffusion.py:209-255) at SD-v1.5 width and the reference's
training size: batch 8, 64 x 64 images = 8 x 8 latents, num_inference_steps = 5 -> one V5 pass per step (ldiffusion.py:198), contrastive
loss, backward through the frozen VAE decoder and the UNet, AdamW on 859.5 M + 0.6 M parameters.  usage: python scripts/bench_train.py [--graph]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import configs, train, weights
from ldiffusion_amd.scheduler import PNDMScheduler

DEV = "cuda:0"
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
unet = train.TrainableUNet(ucfg, usd, DEV)
dec = train.FrozenVAEDecoder(vcfg, vsd, DEV)
del usd, vsd
g = torch.Generator().manual_seed(0)
B = 8
z0 = (torch.randn((B, 4, 8, 8), generator=g) * 0.8).to(DEV)
hidden = (torch.randn((B, 6, 768), generator=g) * 0.5).to(DEV)
proj = ((torch.randn((768, 768), generator=g) / 768 ** 0.5).to(DEV).requires_grad_(True), torch.zeros(768, device=DEV, requires_grad=True))
sch = PNDMScheduler()
sch.set_timesteps(1, device=DEV)
ts = [int(t) for t in sch.timesteps]
eps32 = torch.finfo(torch.float32).eps
u_list = [(torch.rand((B, 4, 8, 8), generator=g) * (2 - eps32) + (eps32 - 1)).to(DEV) for _ in ts]
pairs = [[(int(torch.randint(0, 4096, (1,), generator=g)), int(torch.randint(0, 4096, (1,), generator=g)), torch.randint(0, 4096, (1024,), generator=g).tolist())
          for _ in range(8)] for _ in range(B)]
state = {}
if "--graph" in sys.argv:   # forward + loss + backward replayed from one captured HIP graph (train.GraphedStep)
    gstep = train.GraphedStep(unet, dec, proj, B, ts, sch.alphas_cumprod, latent_hw=8, text_len=6, text_dim=768, max_triples=256, num_negatives=1024)
    step = lambda: train.train_step_graphed(gstep, z0, hidden, u_list, pairs, state, lr=1e-5)
else:
    step = lambda: train.train_step(unet, dec, proj, z0, hidden, ts, sch.alphas_cumprod, u_list, pairs, state, lr=1e-5)
losses = [step() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    losses.append(step())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
nparam = sum(p.numel() for p in unet.parameters()) + sum(p.numel() for p in proj)
print(f"training step{' (graph replay)' if '--graph' in sys.argv else ''}, SD-v1.5 width, B={B}, 8x8 latents, {len(ts)} V5 pass(es): {dt * 1e3:.1f} ms/step = {B / dt:.1f} images/s; {nparam / 1e6:.1f} M trainable parameters; "
      f"loss {losses[0]:.4f} -> {losses[-1]:.4f}; peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
