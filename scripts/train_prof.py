import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from ldiffusion_amd import configs, train, weights, autograd as ag
DEV="cuda:0"
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
unet = train.TrainableUNet(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), DEV)
dec = train.FrozenVAEDecoder(vcfg, weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True), DEV)
g = torch.Generator().manual_seed(0)
z = (torch.randn((8,4,8,8), generator=g)).to(DEV); ctx = (torch.randn((8,6,768), generator=g)*0.5).to(DEV)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
def fwd():
    with torch.no_grad(): return unet(z, 501, ctx)
def fwdbwd():
    for p in unet.parameters(): p.grad=None
    out = dec(unet(z, 501, ctx)); out.square().mean().backward()
state={}
params = unet.parameters()
def adam(): ag.adamw_step(params, [p.grad for p in params], state, 1e-5)
print("unet fwd (no grad):", t(fwd), "ms")
print("unet+dec fwd+bwd:", t(fwdbwd), "ms")
print("adamw:", t(adam), "ms")
print("clip:", t(lambda: train.clip_grad_norm(params, 1.0)), "ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    fwdbwd(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=10, max_name_column_width=60))
