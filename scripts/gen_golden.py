"""Generate tests/golden/*.npz (data only).  Two kinds of fixtures, recorded in the `provenance` field:

  third_party : produced by INSTALLED third-party code the reference's path calls -- these PIN the oracle
      laplace_torch.npz   torch.distributions.Laplace(0, scale).sample()      (ldiffusion.py:235-236)
      luma_pil.npz        PIL Image.convert("L")                              (pixel_latent_vector.py:85)
      round_numpy.npz     (x*255).round().astype(uint8)                       (numpy_to_pil; segmentor.py:107)
  restatement : produced by the oracle itself (diffusers is absent, so nothing can pin these) --
      regression vectors that freeze the restated semantics and feed the GPU parity tests
      pndm.npz            alphas_cumprod, PLMS timestep lists, one PLMS trajectory
      tiny_graph.npz      reduced-width UNet / VAE / 5-pass sampler outputs for seeded inputs and seeded weights
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def laplace_torch():
    scales, samples, us = [], [], []
    finfo = torch.finfo(torch.float32)
    for seed, scale in [(0, 0.0413), (1, 0.573), (2, 0.851), (3, 0.972)]:  # sqrt(1-abar_t) at t = 1, 251, 501, 751
        torch.manual_seed(seed)
        s = torch.distributions.Laplace(0, torch.tensor(scale)).sample((256,))
        torch.manual_seed(seed)
        u = torch.empty(256).uniform_(finfo.eps - 1, 1)   # the draw rsample() makes internally
        scales.append(scale); samples.append(s.numpy()); us.append(u.numpy())
    np.savez_compressed(os.path.join(OUT, "laplace_torch.npz"), provenance="third_party: torch.distributions.Laplace " + torch.__version__,
                        scale=np.array(scales, np.float32), u=np.stack(us), sample=np.stack(samples))


def luma_pil():
    from PIL import Image
    import PIL
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    rgb[0, :6] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 2, 3]]
    L = np.array(Image.fromarray(rgb).convert("L"))
    np.savez_compressed(os.path.join(OUT, "luma_pil.npz"), provenance="third_party: Pillow " + PIL.__version__, rgb=rgb, luma=L)


def round_numpy():
    x = np.concatenate([np.linspace(0, 1, 1001, dtype=np.float32), (np.arange(256, dtype=np.float32) + 0.5) / 255.0,
                        np.array([0.5 / 255, 1.5 / 255, 2.5 / 255, 0.4999, 0.5, 0.50196078], np.float32)])
    np.savez_compressed(os.path.join(OUT, "round_numpy.npz"), provenance="third_party: numpy " + np.__version__, x=x,
                        u8=(x * 255).round().astype("uint8"))


def pndm():
    from oracle import schedule
    sch = schedule.PNDMOracle()
    g = torch.Generator().manual_seed(0)
    out = dict(provenance="restatement: oracle/schedule.py (diffusers PNDMScheduler semantics, unpinned)",
               alphas_cumprod=sch.alphas_cumprod.numpy())
    for n in (1, 2, 4, 5, 10, 19, 20):
        out[f"timesteps_{n}"] = schedule.plms_timesteps(n)
    sch.set_timesteps(4)
    x = torch.randn((1, 4, 4, 4), generator=g)
    traj, epss = [x.numpy().copy()], []
    for t in sch.timesteps:
        eps = torch.randn((1, 4, 4, 4), generator=g)
        x = sch.step(eps, t, x).prev_sample
        epss.append(eps.numpy()); traj.append(x.numpy().copy())
    out["traj_eps"], out["traj_x"] = np.stack(epss), np.stack(traj)
    np.savez_compressed(os.path.join(OUT, "pndm.npz"), **out)


def tiny_graph():
    from ldiffusion_amd import configs, weights
    from oracle import pipeline as op
    ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43)
    pipe = op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg))
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((2, 3, 64, 64), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    lat_in = torch.randn((2, 4, 8, 8), generator=g)
    eps = pipe.unet(lat_in, 501, ctx).sample
    mom = torch.cat([pipe.vae.encode(x).latent_dist.mean, pipe.vae.encode(x).latent_dist.logvar], 1)
    dec = pipe.vae.decode(lat_in).sample
    s5 = op.sample_v6(pipe, x, ctx, 5)
    np.savez_compressed(os.path.join(OUT, "tiny_graph.npz"),
                        provenance="restatement: oracle/{unet,vae,pipeline}.py with weights.synthetic_state_dict(seed 42/43), torch " + torch.__version__,
                        images=x.numpy(), ctx=ctx.numpy(), lat_in=lat_in.numpy(), unet_eps_t501=eps.numpy(), vae_moments=mom.numpy(),
                        vae_decode=dec.numpy(), v6_latents=np.stack([l.numpy() for l in s5["latents"]]), v6_features=s5["features"],
                        v6_rgb_last=s5["rgb_u8"][:, -1], v6_unet_calls=np.array(pipe.unet.calls[-5:]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    laplace_torch(); luma_pil(); round_numpy(); pndm(); tiny_graph()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
