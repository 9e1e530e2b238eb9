"""Where one fine-tuning step spends its time (wall clock with synchronisation after each phase; scripts/bench_train.py's workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ldiffusion_amd import configs, train, weights, autograd as ag
from ldiffusion_amd.scheduler import PNDMScheduler
DEV = "cuda:0"
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
unet = train.TrainableUNet(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), DEV)
dec = train.FrozenVAEDecoder(vcfg, weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True), DEV)
g = torch.Generator().manual_seed(0)
B = 8
z0 = (torch.randn((B, 4, 8, 8), generator=g) * 0.8).to(DEV)
hidden = (torch.randn((B, 6, 768), generator=g) * 0.5).to(DEV)
proj = ((torch.randn((768, 768), generator=g) / 768 ** 0.5).to(DEV).requires_grad_(True), torch.zeros(768, device=DEV, requires_grad=True))
sch = PNDMScheduler(); sch.set_timesteps(1, device=DEV); ts = [int(t) for t in sch.timesteps]
eps32 = torch.finfo(torch.float32).eps
u_list = [(torch.rand((B, 4, 8, 8), generator=g) * (2 - eps32) + (eps32 - 1)).to(DEV) for _ in ts]
pairs = [[(int(torch.randint(0, 4096, (1,), generator=g)), int(torch.randint(0, 4096, (1,), generator=g)), torch.randint(0, 4096, (1024,), generator=g).tolist())
          for _ in range(8)] for _ in range(B)]
state = {}
params = unet.parameters() + list(proj)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for it in range(4):
    for p in params: p.grad = None
    t0 = sync()
    ctx = F.linear(hidden, proj[0], proj[1])
    feats, rgb = train.v5_features(unet, dec, z0, ctx, ts, sch.alphas_cumprod, u_list)
    t1 = sync()
    loss = train.contrastive_loss(feats, pairs)
    t2 = sync()
    loss.backward()
    t3 = sync()
    train.clip_grad_norm(params, 1.0)
    t4 = sync()
    ag.adamw_step(params, [p.grad for p in params], state, lr=1e-5)
    t5 = sync()
    if it >= 1:
        for k, v in (("forward (UNet + frozen decoder + features)", t1 - t0), ("contrastive loss", t2 - t1), ("backward", t3 - t2), ("clip", t4 - t3), ("AdamW", t5 - t4)):
            acc[k] = acc.get(k, 0.0) + v / 3
for k, v in acc.items(): print(f"{k:45s} {v*1e3:7.1f} ms")
print(f"{'total':45s} {sum(acc.values())*1e3:7.1f} ms")
