"""A/B of the bench step loop (every step joins its own decodes) against a two-deep pipeline: two samplers over the same UNet / VAE
handles in overlap mode 2 (deferred join), batch k+1 enqueued before batch k is joined and finished, so the trailing decodes of batch k
run beside the encoder and the first UNet passes of batch k+1.  Same work, same results; diagnostic.  usage: python scripts/bench_pipelined.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline, argmax_mask
dev = torch.device("cuda:0")
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, dev), UNet2DConditionModel(ucfg, usd, dev))
s1, s2 = LaplaceSampler(pipe), LaplaceSampler(pipe)
g = torch.Generator().manual_seed(1234)
images = torch.rand((8, 3, 512, 512), generator=g).to(dev)
ctx = (torch.randn((1, 6, 768), generator=torch.Generator().manual_seed(1235)) * 0.5).to(dev)
hg = torch.Generator().manual_seed(1236)
head_w = (torch.randn((6, 5), generator=hg) / 5 ** 0.5).to(dev)
head_b = (0.1 * torch.randn(6, generator=hg)).to(dev)
def finish(out):
    logits = torch.einsum("cn,bnhw->bchw", head_w, out["features"].float() * (1.0 / 255.0)) + head_b[None, :, None, None]
    return argmax_mask(logits)
def serial(n):
    s1.set_overlap(1)
    m = None
    for _ in range(n):
        m = finish(s1.sample(images, ctx, 5))
    return m
def pipelined(n):
    s1.set_overlap(2); s2.set_overlap(2)
    prev = None
    m = None
    for i in range(n):
        s = (s1, s2)[i & 1]
        out = s.sample(images, ctx, 5)
        if prev is not None:
            prev[0].join(); m = finish(prev[1])
        prev = (s, out)
    prev[0].join()
    return finish(prev[1])
def timeit(fn, n):
    fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); m = fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, m
for rep in range(2):
    a, ma = timeit(serial, 6)
    b, mb = timeit(pipelined, 6)
    print(f"serial {a:.2f} ms/step   pipelined {b:.2f} ms/step   ({(a / b - 1) * 100:+.2f} %)   masks equal: {bool(torch.equal(ma, mb))}")
