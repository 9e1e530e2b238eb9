"""Experiment (round 6): the bench's two batches in flight as TWO INDEPENDENT PIPELINES (own UNet / VAE handles = own workspaces, own main stream and decode
side stream each) against the product's form (one pair of handles, two samplers with deferred joins: UNet chains serialised on one stream).
With two pipelines the deep UNet levels of one batch (launches that leave most CUs idle) can run beside the other batch's UNet / encoder launches as well as
beside decodes.  Same work, same per-batch results.  usage: python scripts/bench_two_pipelines.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline, probe_argmax_mask
dev = torch.device("cuda:0")
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
pipes = [StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, dev), UNet2DConditionModel(ucfg, usd, dev)) for _ in range(2)]
g = torch.Generator().manual_seed(1234)
images = torch.rand((8, 3, 512, 512), generator=g).to(dev)
ctx = (torch.randn((1, 6, 768), generator=torch.Generator().manual_seed(1235)) * 0.5).to(dev)
hg = torch.Generator().manual_seed(1236)
head_w = (torch.randn((6, 5), generator=hg) / 5 ** 0.5).to(dev)
head_b = (0.1 * torch.randn(6, generator=hg)).to(dev)
finish = lambda out: probe_argmax_mask(out["features"], head_w, head_b, 1.0 / 255.0)

# product form: one pipeline, two samplers, deferred joins
s1, s2 = LaplaceSampler(pipes[0]), LaplaceSampler(pipes[0])
def product(n):
    s1.set_overlap(2); s2.set_overlap(2)
    prev, m = None, None
    for i in range(n):
        s = (s1, s2)[i & 1]
        out = s.sample(images, ctx, 5)
        if prev is not None:
            prev[0].join(); m = finish(prev[1])
        prev = (s, out)
    prev[0].join()
    m = finish(prev[1])
    s1.set_overlap(1); s2.set_overlap(1)
    return m

# two pipelines, each on a stream of its own, batches dealt alternately; every call joins its own decodes on its own stream
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
samplers = [LaplaceSampler(pipes[0]), LaplaceSampler(pipes[1])]
def two_pipelines(n, mode=1):
    ms = [None, None]
    for sp in samplers: sp.set_overlap(mode)
    for st in streams: st.wait_stream(torch.cuda.current_stream())
    for i in range(n):
        k = i & 1
        with torch.cuda.stream(streams[k]):
            out = samplers[k].sample(images, ctx, 5)
            if mode == 2: samplers[k].join()
            ms[k] = finish(out)
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    for sp in samplers: sp.set_overlap(1)
    return ms[(n - 1) & 1]

def timeit(fn, n):
    fn(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); m = fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, m
for rep in range(2):
    a, ma = timeit(product, 6)
    b, mb = timeit(two_pipelines, 6)
    print(f"one pipeline, two samplers (product) {a:.2f} ms/step   two pipelines on two streams {b:.2f} ms/step   ({(a / b - 1) * 100:+.2f} %)   masks equal: {bool(torch.equal(ma, mb))}", flush=True)
