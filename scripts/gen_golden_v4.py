"""Generate tests/golden/reference_v4.npz by running the REFERENCE's own `utils.copy_or_convert_image` (/root/reference/utils.py:176-208, the
sampler variant V4 that nnU-Net dataset creation calls per image) in THIS container on the oracle's duck-typed objects.  Run once here; only
the captured data is committed, /root/reference is never read at test time.

What is stubbed for the duration of the script (the reference cannot be imported as shipped):
  * third-party modules exactly as scripts/gen_golden_reference.py does (its install_stubs), with a real `transforms.Normalize`;
  * the function hard-codes device = 'cuda' (utils.py:178): `Tensor.to` / `Module.to` map the string 'cuda' to 'cpu' while it runs.
The reference function itself is executed unmodified.  It receives as `unet` what the reference passes at its call site
(segmentor.py:209-217 -> utils.py:210,249): a `_UNetTextAlignWrapper` (segmentor.py:183-205; the class is local to a method there, restated
below with a call record) around the base UNet, holding the cached prompt embeddings.  Because the function projects the text embeddings
with a fresh nn.Linear(768, 1280) (utils.py:193,197), their last dimension never equals the UNet's cross_attention_dim and the wrapper falls
back to the cached embeddings: the fixture pins exactly that (F12), the timestep that reaches the UNet, and the decode -> PIL -> PNG tail.
"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
OUT = os.path.join(ROOT, "tests", "golden")


class AlignWrapper:
    """segmentor.py:183-205 restated (test infrastructure), recording which branch ran and what reached the UNet."""

    def __init__(self, base_unet, default_text_embeddings):
        self.base_unet, self.default_text_embeddings = base_unet, default_text_embeddings
        self.cross_attention_dim = base_unet.config.cross_attention_dim
        self.config = base_unet.config
        self.record = []

    def __call__(self, sample, timestep, encoder_hidden_states, *args, **kwargs):
        use_fallback = encoder_hidden_states is None or encoder_hidden_states.shape[-1] != self.cross_attention_dim
        given = None if encoder_hidden_states is None else list(encoder_hidden_states.shape)
        if use_fallback:
            emb = self.default_text_embeddings
            if emb.shape[0] != sample.shape[0]:
                emb = emb[:1].expand(sample.shape[0], -1, -1)
            encoder_hidden_states = emb
        self.record.append((int(timestep), bool(use_fallback), given, list(sample.shape)))
        return self.base_unet(sample, timestep, encoder_hidden_states.to(dtype=torch.float32), *args, **kwargs)


def main():
    import gen_golden_reference as G
    G.install_stubs()
    tr = sys.modules["torchvision.transforms"]

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tr.Normalize = Normalize
    t_to, m_to = torch.Tensor.to, torch.nn.Module.to
    fix = lambda a: tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
    torch.Tensor.to = lambda self, *a, **k: t_to(self, *fix(a), **{kk: ("cpu" if isinstance(v, str) and v.startswith("cuda") else v) for kk, v in k.items()})
    torch.nn.Module.to = lambda self, *a, **k: m_to(self, *fix(a), **k)
    try:
        from PIL import Image
        from ldiffusion_amd import configs, weights
        from oracle import pipeline as op
        utils = importlib.import_module("LDiffusion.utils")
        ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
        usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
        vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43)
        unet, vae = op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg)
        pipe = op.OraclePipeline(unet, vae, G.FakeTokenizer(), G.FakeTextEncoder(768))   # hidden 768: the function builds nn.Linear(768, 1280)
        cached = torch.randn((1, 5, ucfg["cross_attention_dim"]), generator=torch.Generator().manual_seed(11)) * 0.5   # what _get_text_embeddings cached
        wrapped = AlignWrapper(unet, cached)
        img = Image.fromarray((torch.rand((80, 96, 3), generator=torch.Generator().manual_seed(21)) * 255).to(torch.uint8).numpy(), "RGB")
        with tempfile.TemporaryDirectory() as td:
            src, dst, dst2 = os.path.join(td, "in.png"), os.path.join(td, "out.png"), os.path.join(td, "copy.png")
            img.save(src)
            torch.manual_seed(3)   # the nn.Linear(768, 1280) initialisation inside the function (its output is discarded by the wrapper)
            utils.copy_or_convert_image(img, src, dst, pipeline=pipe, unet=wrapped, use_diffusion=True)        # REFERENCE CODE, utils.py:176-206
            out = np.asarray(Image.open(dst).convert("RGB"), np.uint8)
            utils.copy_or_convert_image(img, src, dst2, pipeline=pipe, unet=wrapped, use_diffusion=False)      # ... :207-208: a plain copy
            copied = open(src, "rb").read() == open(dst2, "rb").read()
    finally:
        torch.Tensor.to, torch.nn.Module.to = t_to, m_to
    pooled = torch.nn.functional.avg_pool2d(torch.from_numpy(out).permute(2, 0, 1).float()[None], 64)[0].numpy()
    rec = wrapped.record
    np.savez_compressed(os.path.join(OUT, "reference_v4.npz"), out_shape=np.array(out.shape), pooled=pooled, crop=out[480:544, 480:544].copy(),
                        calls_t=np.array([r[0] for r in rec]), calls_fallback=np.array([r[1] for r in rec]), calls_given_dim=np.array([r[2][-1] for r in rec]),
                        calls_sample_shape=np.array([r[3] for r in rec]), base_unet_calls=np.array(unet.calls), plain_copy=np.array(copied),
                        image_seed=np.array(21), cached_seed=np.array(11), image_hw=np.array([80, 96]))
    print("reference V4: out", out.shape, "wrapper calls", rec, "base UNet timesteps", unet.calls, "plain copy identical:", copied)


if __name__ == "__main__":
    main()
