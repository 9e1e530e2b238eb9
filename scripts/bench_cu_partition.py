"""Does giving the two streams of the pipelined bench step disjoint CU shares beat letting them share the whole chip?  The bench loop
(two samplers, overlap mode 2) with the decode side stream restricted to a share of every XCD's CUs (ldiff_vae_set_side_cu_share) and,
optionally, the main stream (encoder, UNet, PLMS, mask tail) restricted to the complement (ldiff_stream_create_cu_share).  Same work, same
results (masks compared); diagnostic.  usage: python scripts/bench_cu_partition.py [steps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib, configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline, probe_argmax_mask
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
vae = AutoencoderKL(vcfg, vsd, dev)
pipe = StableDiffusionImg2ImgPipeline(vae, UNet2DConditionModel(ucfg, usd, dev))
s1, s2 = LaplaceSampler(pipe), LaplaceSampler(pipe)
g = torch.Generator().manual_seed(1234)
images = torch.rand((8, 3, 512, 512), generator=g).to(dev)
ctx = (torch.randn((1, 6, 768), generator=torch.Generator().manual_seed(1235)) * 0.5).to(dev)
hg = torch.Generator().manual_seed(1236)
head_w = (torch.randn((6, 5), generator=hg) / 5 ** 0.5).to(dev)
head_b = (0.1 * torch.randn(6, generator=hg)).to(dev)
lib = _lib.load()


def finish(out):
    return probe_argmax_mask(out["features"], head_w, head_b, 1.0 / 255.0)


def pipelined(n):
    s1.set_overlap(2); s2.set_overlap(2)
    prev, m = None, None
    for i in range(n):
        s = (s1, s2)[i & 1]
        out = s.sample(images, ctx, 5, want_features=True, want_rgb=True)
        if prev is not None:
            prev[0].join(); m = finish(prev[1])
        prev = (s, out)
    prev[0].join()
    m = finish(prev[1])
    s1.set_overlap(1); s2.set_overlap(1)
    return m


def timeit(n):
    pipelined(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); m = pipelined(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, m


def vae_handle():
    for name in ("handle", "_h", "h"):
        if hasattr(vae, name):
            return getattr(vae, name)
    raise RuntimeError("no VAE handle attribute")


def run(side, main):
    """side / main: (lo32, hi32) or None = whole chip"""
    _lib.check(lib.ldiff_vae_set_side_cu_share(vae_handle(), *(side or (0, 32))))
    if main is None:
        return timeit(steps)
    raw = C.c_void_p()
    _lib.check(lib.ldiff_stream_create_cu_share(main[0], main[1], C.byref(raw)))
    ext = torch.cuda.ExternalStream(raw.value, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(ext):
        r = timeit(steps)
    torch.cuda.synchronize()
    _lib.check(lib.ldiff_stream_destroy(raw))
    return r


base, m0 = run(None, None)
print(f"whole chip for both streams: {base:.2f} ms/step", flush=True)
cases = [((0, 24), None), ((0, 16), None), ((8, 32), (0, 8)), ((0, 24), (24, 32)), ((0, 16), (16, 32)), ((0, 24), (8, 32)), ((0, 16), (8, 32)), (None, None)]
for side, main in cases:
    t, m = run(side, main)
    print(f"side {side}  main {main}: {t:.2f} ms/step ({(base / t - 1) * 100:+.2f} %)  masks equal: {bool(torch.equal(m, m0))}", flush=True)
