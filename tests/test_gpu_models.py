"""-m gpu: UNet / VAE / scheduler / sampler through the python shims (which bind the C ABI) against the CPU oracle.

Tolerances.  BASELINE.json's north star: latents within 1e-3 of the (fp32) CPU path with an fp16 UNet, identical arg-max masks.
All errors are max|got - ref| / max|ref| (relative to the range of the compared tensor).  Checkpoints are fp16-valued (configs[1]:
"fp16 SD-v1.5 UNet") and both sides load the same values; every MFMA operand is fp16, the residual stream is kept as fp16 hi|lo
pairs.  Asserted:
  * sampler latents after every pass, VAE encode mean: <= 1e-3 (measured 0.7e-4 .. 4.7e-4 at SD-v1.5 width, <= 3.6e-4 tiny),
  * one UNet pass (eps): <= 1e-3 at SD-v1.5 width (measured 7.4e-4), <= 1.2e-3 at 1/5 width (measured 6.0e-4 .. 9.1e-4: a
    64-channel graph averages less rounding noise per output than a 320-channel one),
  * VAE decode (image range, not a latent; decoder default = all-fp16 storage): <= 4e-3 (measured 2.0e-3 .. 2.3e-3); uint8 images /
    luma: at most 1 grey level apart,
  * integer kernels given identical inputs (luma, uint8 rounding, argmax): bit exact.
"""
import os

import numpy as np
import pytest
import torch

from ldiffusion_amd import configs, weights
from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
from ldiffusion_amd.pipeline import (LaplaceSampler, StableDiffusionImg2ImgPipeline, argmax_mask, laplace_noise, luma_float)
from ldiffusion_amd.scheduler import PNDMScheduler
from oracle import noise_post, pipeline as op, schedule as osched

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# "Identical arg-max masks" (north star, segmentor.py:536-541) is not attainable bit for bit with fp16 MFMA operands: any float error moves some uint8
# rounding boundaries of the luma features (6-7 % of them end one grey level off with the all-fp16 decoder: DESIGN.md section 2).  What IS asserted is
# the contract that statement leaves: the device mask may differ from the oracle's ONLY at pixels where the oracle itself is within that rounding of a
# tie -- per differing pixel, the oracle's logit margin between its own class c1 and the device's class c2 is at most what the pixel's feature
# differences (each asserted <= one grey level) can move it:  lg[c1] - lg[c2] <= sum_n |w[c1,n] - w[c2,n]| * |f_dev[n] - f_ref[n]|
# (<= sum_n |w[c1,n] - w[c2,n]|, the one-grey-level-on-every-feature bound).
# That contract says WHERE a mask may differ, not how often: since the device mask is the arg-max of the same probe over device features that are within
# one grey level, a kernel change that multiplied the off-by-one rate would still satisfy it.  So every real-size test ALSO bounds the count, at a few
# times the measured value (rounds 4-5: 0-11 of 524,288 pixels at configs[1], 200 of 262,144 = 99.92 % agreement at configs[3]): a jump is a regression.
MASK_FLIP_BOUND = 32        # configs[1]: differing pixels of 524,288 (two 512 x 512 masks)
MASK_AGREE_20_PASSES = 0.997   # configs[3] / config 4: 20 uint8 features per pixel


def assert_mask_flips_within_margin(mask, rmask, ref_features, W, bias, dev_features=None, what=""):
    """mask / rmask uint8 [B,H,W] (device / oracle); ref_features uint8 [B,N,H,W] (the oracle's); W [C,N] and bias [C] in logit units per RAW grey
    level (the probe applied to features 0..255).  dev_features (same shape) tightens the bound to the features that really differ; without it the
    bound is one grey level on every feature (merged / averaged logits).  Returns the number of differing pixels."""
    mask, rmask = np.asarray(mask), np.asarray(rmask)
    idx = np.argwhere(mask != rmask)
    if len(idx) == 0:
        return 0
    b, y, x = idx.T
    W64, b64 = W.double().numpy(), bias.double().numpy()
    f = np.asarray(ref_features)[b, :, y, x].astype(np.float64)                      # [K, N]
    lg = f @ W64.T + b64                                                              # oracle logits at the differing pixels, float64
    k = np.arange(len(idx))
    c1, c2 = rmask[b, y, x].astype(int), mask[b, y, x].astype(int)
    margin = lg[k, c1] - lg[k, c2]
    dW = np.abs(W64[c1] - W64[c2])                                                    # [K, N]
    if dev_features is not None:
        df = np.abs(np.asarray(dev_features)[b, :, y, x].astype(np.float64) - f)
        assert df.max() <= 1
        bound = (dW * df).sum(1)
    else:
        bound = dW.sum(1)
    slack = 4e-6 * (np.abs(lg).max() + 1.0)                                           # float32 accumulation of the probe on both sides
    worst = float((margin - bound).max())
    print(f"  mask contract{' ' + what if what else ''}: {len(idx)} of {mask.size} pixels differ; oracle margin at those pixels <= {margin.max():.4f}, "
          f"allowed by their feature differences: worst (margin - bound) {worst:.2e}")
    assert (margin >= -slack).all(), "the oracle's own class is not its arg-max"
    assert worst <= slack, (f"north star (segmentor.py:536-541): a mask pixel differs from the oracle's where the oracle is NOT within one grey level "
                            f"of a tie: margin - bound = {worst:.3e}")
    return len(idx)


def rel_err(got, ref):
    """max|got - ref| / max|ref|: the error relative to the RANGE of the compared tensor.  This is the reading of BASELINE.json's "within 1e-3
    on latents" that the tests assert (latents reach +-10..20 after a few passes, so 1e-3 of range is ~1e-2 absolute); `err_report` prints
    the absolute maximum and the RMS error beside it wherever the contract is checked."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert torch.isfinite(got).all(), "non-finite output"
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()


def err_report(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    d = got - ref
    return (f"max|err| {d.abs().max().item():.3e} abs = {(d.abs().max() / ref.abs().max().clamp_min(1e-6)).item():.3e} of range (max|ref| {ref.abs().max().item():.2f}), "
            f"rms err {d.pow(2).mean().sqrt().item():.3e} = {(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12)).item():.3e} of rms(ref)")


@pytest.fixture(scope="module")
def tiny():
    ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    unet, vae = UNet2DConditionModel(ucfg, usd, DEV), AutoencoderKL(vcfg, vsd, DEV)
    pipe = StableDiffusionImg2ImgPipeline(vae, unet)
    opipe = op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg))
    return dict(ucfg=ucfg, vcfg=vcfg, usd=usd, vsd=vsd, unet=unet, vae=vae, pipe=pipe, opipe=opipe)


@pytest.mark.parametrize("B,h,w,L,t", [(2, 16, 16, 6, 751), (1, 8, 24, 77, 1), (3, 32, 32, 1, 501)])
def test_unet_forward_tiny(tiny, B, h, w, L, t):
    g = torch.Generator().manual_seed(B * 100 + L)
    x = torch.randn((B, 4, h, w), generator=g)
    ctx = torch.randn((1, L, tiny["ucfg"]["cross_attention_dim"]), generator=g) * 0.5
    out = tiny["unet"](x.to(DEV), torch.tensor(t), ctx.to(DEV))
    ref = tiny["opipe"].unet(x, t, ctx)
    e = rel_err(out.sample, ref.sample)
    print(f"unet tiny B={B} {h}x{w} L={L} t={t}: rel err {e:.3e}")
    assert out[0] is out.sample and e <= 1.2e-3


def test_unet_per_sample_context_and_errors(tiny):
    g = torch.Generator().manual_seed(9)
    x = torch.randn((2, 4, 16, 16), generator=g)
    ctx = torch.randn((2, 5, 64), generator=g)
    out = tiny["unet"](x.to(DEV), 251, ctx.to(DEV)).sample
    ref = tiny["opipe"].unet(x, 251, ctx).sample
    assert rel_err(out, ref) <= 1.2e-3
    with pytest.raises(ValueError):
        tiny["unet"](x.to(DEV), 1, torch.randn((3, 5, 64), device=DEV))       # context batch mismatch
    with pytest.raises(ValueError):
        tiny["unet"](x.to(DEV), 1, torch.randn((1, 5, 32), device=DEV))       # wrong cross_attention_dim
    with pytest.raises(ValueError):
        tiny["unet"](torch.randn((2, 4, 12, 12), device=DEV), 1, ctx.to(DEV))  # latent size not divisible by 8
    with pytest.raises(ValueError):
        tiny["unet"](torch.randn((2, 3, 16, 16), device=DEV), 1, ctx.to(DEV))  # wrong channel count


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 32, 96)])
def test_vae_encode_decode_tiny(tiny, B, H, W):
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand((B, 3, H, W), generator=g)
    dist = tiny["vae"].encode(x.to(DEV)).latent_dist
    odist = tiny["opipe"].vae.encode(x).latent_dist
    e_mean = rel_err(dist.mean, odist.mean)
    e_logvar = rel_err(dist.logvar, odist.logvar)
    z = odist.mean
    dec = tiny["vae"].decode(z.to(DEV)).sample
    odec = tiny["opipe"].vae.decode(z).sample
    e_dec = rel_err(dec, odec)
    print(f"vae tiny {B}x{H}x{W}: mean {e_mean:.3e} logvar {e_logvar:.3e} decode {e_dec:.3e}")
    assert e_mean <= 3e-4 and e_logvar <= 3e-4 and e_dec <= 4e-3
    assert dist.sample().shape == dist.mean.shape


def test_fp16_overflow_in_the_decoder_is_detected(tiny):
    """include/ldiff.h "Non-finite detection": the reference decodes z / 0.18215 of UN-scaled latents in fp32 (pixel_latent_vector.py:73,81); here every
    activation is stored as fp16, so a checkpoint whose activations leave +-65504 must raise instead of returning a plausible-looking image / mask.
    The synthetic decoder's mid_block.resnets.0.conv1 is scaled by powers of two until its fp32 output passes 65504 -- its output only feeds norm2,
    which is scale-invariant, so the fp32 oracle's image is finite and nearly unchanged while the fp16 graph overflows."""
    import torch.nn.functional as F
    from ldiffusion_amd._lib import NonFiniteError
    from oracle.unet import _conv, _gn
    from oracle.vae import VAE_EPS, vae_decode
    vcfg, vsd = tiny["vcfg"], tiny["vsd"]
    g = torch.Generator().manual_seed(11)
    z = torch.randn((2, 4, 8, 8), generator=g) * 0.3
    key = "decoder.mid_block.resnets.0.conv1"

    def conv1_absmax(sd):
        h = _conv(sd, "decoder.conv_in", _conv(sd, "post_quant_conv", z / 0.18215, padding=0))
        a = F.silu(_gn(sd, "decoder.mid_block.resnets.0.norm1", h, vcfg["norm_num_groups"], VAE_EPS))
        return _conv(sd, key, a).abs().max().item()

    s, m0 = 1.0, conv1_absmax(vsd)
    while s * m0 <= 2 * 65504:
        s *= 2.0
    bad = dict(vsd)
    bad[key + ".weight"], bad[key + ".bias"] = vsd[key + ".weight"] * s, vsd[key + ".bias"] * s
    assert bad[key + ".weight"].abs().max() < 65504 and conv1_absmax(bad) > 65504
    oimg = vae_decode(bad, vcfg, z / 0.18215)
    assert torch.isfinite(oimg).all(), "the fp32 graph itself must survive (GroupNorm removes the scale)"
    print(f"overflow test: conv1 max|activation| {m0:.3g} x {s:g} = {m0 * s:.4g} > 65504; fp32 oracle finite, max|image change| "
          f"{(oimg - vae_decode(vsd, vcfg, z / 0.18215)).abs().max().item():.2e}")

    good = tiny["vae"]
    good._decode(z.to(DEV), 1 / 0.18215, want_image=True)
    good.check_finite()                                            # (i') a healthy decoder never trips the detector
    vae = AutoencoderKL(vcfg, bad, DEV)
    for mode in (0, 1, 2):                                         # precision modes 1 / 2 add mantissa bits, not range: the hi half of a split tensor is an fp16
        vae.set_precision(2, mode)
        vae._decode(z.to(DEV), 1 / 0.18215, want_image=True)
        with pytest.raises(NonFiniteError, match="65504"):
            vae.check_finite()                                     # (i) the flag fires ...
        vae.check_finite()                                         # ... is reported once and cleared
    vae.set_precision(2, 0)
    # reporting rule (b): without check_finite, the NEXT call on the handle reports the completed, flagged work at entry
    vae._decode(z.to(DEV), 1 / 0.18215, want_image=True)
    torch.cuda.synchronize()
    with pytest.raises(NonFiniteError):
        vae._decode(z.to(DEV), 1 / 0.18215, want_image=True)
    vae.check_finite()
    # the shims that hand results to the host check for the caller
    with pytest.raises(NonFiniteError):
        StableDiffusionImg2ImgPipeline(vae, tiny["unet"]).decode_latents(z.to(DEV))
    sampler = LaplaceSampler(StableDiffusionImg2ImgPipeline(vae, tiny["unet"]))
    x = torch.rand((2, 3, 64, 64), generator=g)
    ctx = torch.randn((1, 6, tiny["ucfg"]["cross_attention_dim"]), generator=g) * 0.5
    sampler.sample(x.to(DEV), ctx.to(DEV), 5)
    with pytest.raises(NonFiniteError, match="VAE"):
        sampler.check_finite()
    sampler.check_finite()
    # and an encoder input far outside fp16's range trips the encoder's flag
    vae.encode(torch.full((1, 3, 64, 64), 3.0e6, device=DEV))
    with pytest.raises(NonFiniteError):
        vae.check_finite()
    # the UNet: a latent of 1e7 overflows conv_in's fp16 output
    tiny["unet"](torch.full((1, 4, 16, 16), 1.0e7, device=DEV), 1, ctx.to(DEV))
    with pytest.raises(NonFiniteError):
        tiny["unet"].check_finite()
    tiny["unet"](torch.randn((1, 4, 16, 16), device=DEV), 1, ctx.to(DEV))
    tiny["unet"].check_finite()


def test_decode_latents_uint8_and_luma(tiny):
    g = torch.Generator().manual_seed(3)
    z = torch.randn((2, 4, 8, 8), generator=g) * 0.3
    img = tiny["pipe"].decode_latents(z.to(DEV))
    oimg = tiny["opipe"].decode_latents(z)
    assert img.shape == oimg.shape == (2, 64, 64, 3) and img.dtype == np.float32
    assert np.abs(img - oimg).max() <= 4e-3
    # fused uint8 + luma slots vs the oracle's integer pipeline applied to the *device* float image: bit exact
    luma = torch.zeros((2, 3, 64, 64), dtype=torch.uint8, device=DEV)
    _, image, rgb = tiny["vae"]._decode(z.to(DEV), 1 / 0.18215, want_image=True, want_rgb=True, luma=luma, slot=1)
    u8 = noise_post.to_uint8(image.cpu().numpy())
    assert np.array_equal(rgb.cpu().numpy(), u8)
    assert np.array_equal(luma[:, 1].cpu().numpy(), noise_post.luma_u8(u8))
    assert (luma[:, 0] == 0).all() and (luma[:, 2] == 0).all()
    pil = tiny["pipe"].numpy_to_pil(img)
    assert pil[0].size == (64, 64) and np.array_equal(np.array(pil[0]), noise_post.to_uint8(img)[0])


def test_scheduler_matches_oracle_step_by_step():
    sch, osch = PNDMScheduler(), osched.PNDMOracle()
    assert torch.equal(sch.alphas_cumprod, osch.alphas_cumprod) and len(sch.alphas_cumprod) == 1000
    g = torch.Generator().manual_seed(0)
    for n in (1, 4, 9, 19):
        sch.set_timesteps(n, device=DEV)
        osch.set_timesteps(n)
        assert sch.timesteps.cpu().tolist() == osch.timesteps.tolist()
        x = torch.randn((2, 4, 8, 8), generator=g)
        xo = x.clone()
        xd = x.to(DEV)
        for t in sch.timesteps:
            assert t.dim() == 0
            eps = torch.randn((2, 4, 8, 8), generator=g)
            assert sch.scale_model_input(xd, t) is xd
            xd = sch.step(eps.to(DEV), t, xd).prev_sample
            xo = osch.step(eps, t, xo).prev_sample
            assert (xd.cpu() - xo).abs().max() <= 2e-6 * xo.abs().max()
    with pytest.raises(ZeroDivisionError):
        sch.set_timesteps(0)


@pytest.mark.parametrize("N", [1, 3, 5])
def test_fused_sampler_matches_oracle_tiny(tiny, N):
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((2, 3, 64, 64), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    s = LaplaceSampler(tiny["pipe"])
    out = s.sample(x.to(DEV), ctx.to(DEV), N)
    if N == 1:
        ref1 = op.sample_one_pass(tiny["opipe"], x, ctx)
        ref = dict(latents=[ref1["latents"]], rgb_u8=ref1["rgb_u8"][:, None], features=noise_post.luma_u8(ref1["rgb_u8"])[:, None])
    else:
        ref = op.sample_v6(tiny["opipe"], x, ctx, N)
    assert s.timesteps(N) == tiny["opipe"].unet.calls[-N:]
    e = rel_err(out["latents"], ref["latents"][-1])
    fd = np.abs(out["features"].cpu().numpy().astype(int) - ref["features"].astype(int))
    rd = np.abs(out["rgb"].cpu().numpy().astype(int) - ref["rgb_u8"][:, -1].astype(int))
    print(f"sampler N={N}: latents rel err {e:.3e}; luma max diff {fd.max()} (>1: {(fd > 1).mean():.4f}); rgb max diff {rd.max()}")
    assert out["features"].shape == (2, N, 64, 64)
    assert e <= 1e-3 and fd.max() <= 1 and rd.max() <= 1
    # luma of the last pass must be the integer luma of the last-pass rgb, bit exact
    assert np.array_equal(out["features"][:, -1].cpu().numpy(), noise_post.luma_u8(out["rgb"].cpu().numpy()))


def test_fused_sampler_equals_stepwise_shims(tiny):
    """ldiff_sample == the reference's loop body driven with the shim objects (same kernels, same order): bit exact."""
    g = torch.Generator().manual_seed(77)
    x = torch.rand((1, 3, 64, 64), generator=g).to(DEV)
    ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
    pipe = tiny["pipe"]
    fused = LaplaceSampler(pipe).sample(x, ctx, 5)
    latents = pipe.vae.encode(x).latent_dist.mean
    pipe.scheduler.set_timesteps(4, device=DEV)
    for t in pipe.scheduler.timesteps:
        latents = pipe.scheduler.scale_model_input(latents, t)
        o = pipe.unet(latents, t, ctx)
        latents = pipe.scheduler.step(o[0], t, latents).prev_sample
    assert (fused["latents"] - latents).abs().max() <= 1e-5 * latents.abs().max()


def test_sampler_rejects_bad_arguments(tiny):
    s = LaplaceSampler(tiny["pipe"])
    x = torch.rand((1, 3, 64, 64), device=DEV)
    ctx = torch.randn((1, 6, 64), device=DEV)
    with pytest.raises(ValueError):
        s.sample(x, ctx, 2)      # N=2: set_timesteps(1) yields a single pass (pixel_latent_vector.py:74)
    with pytest.raises(ValueError):
        s.sample(x, ctx, 0)
    with pytest.raises(ValueError):
        s.sample(torch.rand((1, 3, 60, 64), device=DEV), ctx, 3)   # not a multiple of 8
    with pytest.raises(ValueError):
        s.sample(torch.rand((1, 1, 64, 64), device=DEV), ctx, 3)


def test_laplace_given_u_matches_oracle_and_torch_distribution():
    g = torch.Generator().manual_seed(5)
    z0 = torch.randn((2, 4, 8, 8), generator=g)
    u = noise_post.laplace_uniform_draw(z0.shape, g)
    u.view(-1)[0] = 1.1920929e-07 - 1.0  # extreme tail of the open interval
    u.view(-1)[1] = 0.0
    abar = osched.alphas_cumprod()[501]
    ref = noise_post.laplace_forward_noise(z0, abar, u)
    got = laplace_noise(z0.to(DEV), float(torch.sqrt(1 - abar)), u).cpu()
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max()
    # device RNG path: right distribution (mean 0, E|x| = scale, var = 2 scale^2) and counter-based reproducibility
    z = torch.zeros((1, 4, 256, 256), device=DEV)
    a = laplace_noise(z, 0.5, seed=7, offset=0)
    b = laplace_noise(z, 0.5, seed=7, offset=0)
    c = laplace_noise(z, 0.5, seed=8, offset=0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.mean().item()) < 5e-3 and abs(a.abs().mean().item() - 0.5) < 5e-3 and abs(a.var().item() - 0.5) < 2e-2


def test_argmax_bit_exact_and_float_luma():
    g = torch.Generator().manual_seed(11)
    logits = torch.randn((2, 6, 33, 47), generator=g)
    logits[0, :, 0, 0] = 1.0           # all tie -> lowest index
    logits[0, 3, 0, 1] = float("nan")  # NaN counts as the maximum in torch.argmax
    m = argmax_mask(logits.to(DEV)).cpu().numpy()
    assert np.array_equal(m, noise_post.argmax_mask(logits))
    rgb = torch.rand((2, 3, 17, 19), generator=g) * 2 - 0.5
    # float luma: same three products; torch's CPU reduction order over the 3 channels is ISA dependent -> 1-2 ulp
    assert (luma_float(rgb.to(DEV)).cpu() - noise_post.luma_float(rgb)).abs().max() <= 2.5e-7
    assert argmax_mask(torch.zeros((0, 6, 4, 4), device=DEV)).shape == (0, 4, 4)


def test_checkpoint_roundtrip_diffusers_layout(tiny, tmp_path):
    """save_pretrained / from_pretrained keep the diffusers directory layout the reference reads (ldiffusion.py:273, segmentor.py:79)."""
    d = tmp_path / "unet"
    tiny["unet"].save_pretrained(str(d))
    assert (d / "config.json").exists() and (d / "diffusion_pytorch_model.safetensors").exists()
    u2 = UNet2DConditionModel.from_pretrained(str(d), device=DEV)
    assert u2.config.cross_attention_dim == tiny["ucfg"]["cross_attention_dim"]
    x = torch.randn((1, 4, 16, 16), device=DEV)
    ctx = torch.randn((1, 6, 64), device=DEV)
    assert torch.equal(u2(x, 1, ctx).sample, tiny["unet"](x, 1, ctx).sample)
    sd = dict(tiny["usd"])
    sd.pop("conv_in.weight")
    with pytest.raises(RuntimeError):
        UNet2DConditionModel(tiny["ucfg"], sd, DEV)            # missing tensor is an error, not a silent default
    sd = dict(tiny["usd"])
    sd["conv_in.weight"] = torch.zeros((64, 4, 1, 1))
    with pytest.raises(ValueError):
        UNet2DConditionModel(tiny["ucfg"], sd, DEV)            # wrong shape


def test_vae_accepts_deprecated_attention_names(tiny):
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    sd = {}
    for k, v in tiny["vsd"].items():
        for new, old in ren.items():
            if ".attentions.0." + new + "." in k:
                k = k.replace("." + new + ".", "." + old + ".")
        sd[k] = v
    assert any(".query." in k for k in sd)
    v2 = AutoencoderKL(tiny["vcfg"], sd, DEV)
    z = torch.randn((1, 4, 8, 8), device=DEV)
    assert torch.equal(v2.decode(z).sample, tiny["vae"].decode(z).sample)


@pytest.mark.timeout(1500)
def test_sd15_width_unet_and_vae_against_oracle():
    """Full SD-v1.5 widths (859.5 M / 83.7 M params, synthetic weights), B=1, 256x256 patch (32x32 latents)."""
    ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((1, 4, 32, 32), generator=g)
    ctx = torch.randn((1, 6, 768), generator=g) * 0.5
    unet = UNet2DConditionModel(ucfg, usd, DEV)
    out = unet(x.to(DEV), 501, ctx.to(DEV)).sample
    ref = op.OracleUNet(usd, ucfg)(x, 501, ctx).sample
    e_u = rel_err(out, ref)
    del unet, usd
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    vae = AutoencoderKL(vcfg, vsd, DEV)
    img = torch.rand((1, 3, 256, 256), generator=g)
    ovae = op.OracleVAE(vsd, vcfg)
    e_e = rel_err(vae.encode(img.to(DEV)).latent_dist.mean, ovae.encode(img).latent_dist.mean)
    z = torch.randn((1, 4, 32, 32), generator=g)
    e_d = rel_err(vae.decode(z.to(DEV)).sample, ovae.decode(z).sample)
    print(f"SD15 widths: unet rel err {e_u:.3e}, vae encode {e_e:.3e}, vae decode {e_d:.3e}")
    assert e_u <= 1e-3 and e_e <= 3e-4 and e_d <= 4e-3


@pytest.mark.timeout(900)
def test_unet_pass_is_bit_identical_with_and_without_the_dataflow_kernels():
    """The executors send a 1x1 conv / linear to gemm_df_kernel or gemm_dma_kernel by its row count (and the VAE decoder's shortcut convs into the
    dataflow conv or not): the kernels are built to add in the same order, so one SD-v1.5-width UNet pass must come out BIT FOR BIT the same with
    LDIFF_GEMM_DF=0 and =1 (the switch is read once per process: two child processes, CRC32 of the fp32 output)."""
    import subprocess
    import sys
    code = (
        "import sys, zlib, torch; sys.path.insert(0, %r)\n"
        "from ldiffusion_amd import configs, weights\n"
        "from ldiffusion_amd.models import UNet2DConditionModel\n"
        "ucfg = configs.SD15_UNET\n"
        "unet = UNet2DConditionModel(ucfg, weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True), 'cuda:0')\n"
        "g = torch.Generator().manual_seed(5)\n"
        "x = torch.randn((2, 4, 64, 64), generator=g).cuda(); ctx = (torch.randn((1, 6, 768), generator=g) * 0.5).cuda()\n"
        "y = unet(x, 501, ctx).sample; torch.cuda.synchronize()\n"
        "assert torch.isfinite(y).all()\n"
        "print('CRC', zlib.crc32(y.cpu().numpy().tobytes()))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    crcs = []
    for v in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LDIFF_GEMM_DF=v), capture_output=True, text=True, timeout=400)
        assert out.returncode == 0, out.stderr[-2000:]
        crcs.append([l for l in out.stdout.splitlines() if l.startswith("CRC")][-1])
    assert crcs[0] == crcs[1], f"UNet output differs between LDIFF_GEMM_DF=0 and 1: {crcs}"


@pytest.mark.timeout(2400)
def test_config1_sd15_width_512_five_passes_against_oracle():
    """BASELINE.json configs[1] at its real size: SD-v1.5-width UNet + VAE (fp16 checkpoint values), 512x512 patches, the 5-pass
    PLMS sampler, against the fp32 CPU oracle (~20-60 s per patch on the box's cores).
    North-star tolerance: latents within 1e-3 of the reference, as max|diff| / max|ref| over the final latents (the latent
    range; values reach ~10-20 after five passes), asserted for every pass; identical arg-max masks of a 6-class probe over the
    per-pixel latent vectors are reported (uint8 luma features can flip a rounding boundary under ANY float error)."""
    ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    g = torch.Generator().manual_seed(1234)
    B, N = 2, 5
    x = torch.rand((B, 3, 512, 512), generator=g)
    ctx = torch.randn((1, 6, 768), generator=g) * 0.5
    pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, DEV), UNet2DConditionModel(ucfg, usd, DEV))
    out = LaplaceSampler(pipe).sample(x.to(DEV), ctx.to(DEV), N)
    torch.cuda.synchronize()
    # the same loop step by step on the shims, to get the per-pass latents
    lat_dev = []
    latents = pipe.vae.encode(x.to(DEV)).latent_dist.mean
    z0_dev = latents.clone()
    pipe.scheduler.set_timesteps(N - 1, device=DEV)
    for t in pipe.scheduler.timesteps:
        latents = pipe.scheduler.step(pipe.unet(latents, t, ctx.to(DEV))[0], t, latents).prev_sample
        lat_dev.append(latents.clone())
    assert (out["latents"] - lat_dev[-1]).abs().max() <= 1e-5 * lat_dev[-1].abs().max()
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    opipe = op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg))
    z0_ref = opipe.vae.encode(x).latent_dist.mean
    ref = op.sample_v6(opipe, x, ctx, N)
    e_enc = rel_err(z0_dev, z0_ref)
    errs = [rel_err(a, b) for a, b in zip(lat_dev, ref["latents"])]
    fd = np.abs(out["features"].cpu().numpy().astype(int) - ref["features"].astype(int))
    W, bias = _probe_head(6, N, 5)
    lg = torch.einsum("cn,bnhw->bchw", W.to(DEV), out["features"].float()) + bias.to(DEV)[None, :, None, None]
    mask = argmax_mask(lg).cpu().numpy()
    rlg = torch.einsum("cn,bnhw->bchw", W, torch.from_numpy(ref["features"]).float()) + bias[None, :, None, None]
    rmask = np.asarray(noise_post.argmax_mask(rlg))
    agree, ndiff = (mask == rmask).mean(), int((mask != rmask).sum())
    print(f"configs[1] SD15 width 512^2 x {N} passes, B={B}: encode rel err {e_enc:.3e}; latents per pass {[f'{e:.2e}' for e in errs]}; "
          f"luma max diff {fd.max()} (!=0: {(fd > 0).mean():.4f}, >1: {(fd > 1).mean():.6f}); mask agreement {agree:.5f} ({ndiff} of {mask.size} pixels differ)")
    print(f"  final latents: {err_report(lat_dev[-1], ref['latents'][-1])}")
    print(f"  encoder mean : {err_report(z0_dev, z0_ref)}")
    assert e_enc <= 1e-3 and max(errs) <= 1e-3, "north-star tolerance: latents within 1e-3 of the reference (relative to the latent range)"
    assert fd.max() <= 1
    # "identical arg-max masks": the masks may differ only where the oracle is within its features' rounding of a tie (the contract at the top of the file)
    assert assert_mask_flips_within_margin(mask, rmask, ref["features"], W, bias, out["features"].cpu().numpy(), "configs[1] B=2") == ndiff
    assert ndiff <= MASK_FLIP_BOUND, f"{ndiff} mask pixels differ from the oracle's (measured 0-3 in rounds 4-5): regression of the off-by-one rate"
    # the decoder's storage policy (default 0) does not touch the latents; what modes 1 / 2 would buy in the uint8 features, for the record
    for dmode in (1, 2):
        pipe.vae.set_precision(2, dmode)
        f2 = LaplaceSampler(pipe).sample(x.to(DEV), ctx.to(DEV), N)["features"]
        d2 = np.abs(f2.cpu().numpy().astype(int) - ref["features"].astype(int))
        m2 = argmax_mask(torch.einsum("cn,bnhw->bchw", W.to(DEV), f2.float()) + bias.to(DEV)[None, :, None, None]).cpu().numpy()
        print(f"  decoder precision {dmode}: luma max diff {d2.max()} (!=0: {(d2 > 0).mean():.4f}); mask agreement {(m2 == rmask).mean():.5f}")


def test_unet_graph_replay_equals_eager(tiny):
    """The forward is captured into a hipGraph on its second use with a given (B, h, w, context): replays must equal the eager
    launches bit for bit for any timestep and any caller buffers, and a new context / shape / checkpoint must not replay stale state."""
    g = torch.Generator().manual_seed(50)
    unet = tiny["unet"]
    xs = [torch.randn((2, 4, 16, 16), generator=g).to(DEV) for _ in range(3)]
    ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
    ctx2 = (torch.randn((1, 9, 64), generator=g) * 0.5).to(DEV)
    unet.set_graph(False)
    ref = {(i, t, c): unet(xs[i], t, cc).sample.clone() for i in range(3) for t in (751, 1) for c, cc in (("a", ctx), ("b", ctx2))}
    unet.set_graph(True)
    r0 = unet.graph_replays
    for rep in range(2):
        for c, cc in (("a", ctx), ("b", ctx2)):
            for i in range(3):
                for t in (751, 1):
                    assert torch.equal(unet(xs[i], t, cc).sample, ref[(i, t, c)]), (rep, c, i, t)
    assert unet.graph_replays - r0 >= 16          # per context: 1 eager + 1 capture(+replay) + replays
    big = torch.randn((1, 4, 24, 8), generator=g).to(DEV)      # another shape in between drops the cached graph, results unchanged
    unet(big, 5, ctx)
    assert torch.equal(unet(xs[0], 751, ctx).sample, ref[(0, 751, "a")])
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):                                 # replay on a non-default stream
        for _ in range(3):
            o = unet(xs[1], 1, ctx).sample
    s2.synchronize()
    assert torch.equal(o, ref[(1, 1, "a")])


def test_unet_controlnet_additional_residuals(tiny):
    """V7's ControlNet inputs (segmentor.py:357-375): down_block_additional_residuals are added to the skip tensors (not to the mid
    block's input), mid_block_additional_residual to the mid block's output; against the oracle's restatement of that forward."""
    g = torch.Generator().manual_seed(60)
    B, h, w = 2, 16, 16
    x = torch.randn((B, 4, h, w), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    shapes = tiny["unet"]._skip_shapes(B, h, w)
    assert len(shapes) == 12 and shapes[0] == (B, 64, 16, 16) and shapes[-1] == (B, 256, 2, 2)
    down = [torch.randn(sh, generator=g) * 0.3 for sh in shapes]
    mid = torch.randn(shapes[-1], generator=g) * 0.3
    plain = tiny["unet"](x.to(DEV), 501, ctx.to(DEV)).sample
    for dr, mr in ((down, mid), (down, None), (None, mid)):
        got = tiny["unet"](x.to(DEV), 501, ctx.to(DEV), down_block_additional_residuals=None if dr is None else [t.to(DEV) for t in dr],
                           mid_block_additional_residual=None if mr is None else mr.to(DEV)).sample
        ref = tiny["opipe"].unet(x, 501, ctx, down_block_additional_residuals=dr, mid_block_additional_residual=mr).sample
        e = rel_err(got, ref)
        print(f"controlnet residuals down={dr is not None} mid={mr is not None}: rel err {e:.3e}")
        assert e <= 1.2e-3 and not torch.equal(got, plain)
    assert torch.equal(tiny["unet"](x.to(DEV), 501, ctx.to(DEV)).sample, plain)     # the inputs are consumed by one forward
    with pytest.raises(ValueError):
        tiny["unet"](x.to(DEV), 501, ctx.to(DEV), down_block_additional_residuals=[t.to(DEV) for t in down[:-1]])


class _Tok:          # unpadded ids of the prompt, as the fixture generator's stand-in returns them (any padding arguments are accepted)
    def __call__(self, prompts, **kw):
        ids = [[49406, 320, 24857, 5471, 49407] for _ in prompts]
        return {"input_ids": torch.tensor(ids) if kw.get("return_tensors") == "pt" else ids}


class _Enc:
    def __init__(self, hidden):
        self.config = __import__("types").SimpleNamespace(hidden_size=hidden)
        self.table = torch.randn((49408, hidden), generator=torch.Generator().manual_seed(99)) * 0.5

    def __call__(self, ids):
        return {"last_hidden_state": self.table.to(ids.device)[ids]}

    def to(self, *a, **k):
        return self

    def eval(self):
        return self


def test_copy_or_convert_image_v4_against_the_reference_fixture(tmp_path):
    """ldiffusion_amd.utils.copy_or_convert_image (mirror of utils.py:176-208, sampler variant V4) against what the REFERENCE's own function
    wrote when run on the oracle's objects (tests/golden/reference_v4.npz, scripts/gen_golden_v4.py): same image, same cached embeddings
    behind the text-alignment wrapper (the function's own 768 -> 1280 projection is discarded by it, F12), fp32 checkpoint values as there.
    The PNG must agree within one grey level (fp16 operands against fp32) and in its 64 x 64 block means; `use_diffusion=False` copies."""
    from PIL import Image
    from ldiffusion_amd import utils as U
    from ldiffusion_amd.segmentor import TextAlignedUNet
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_v4.npz"))
    ucfg, vcfg = configs.TINY_UNET, configs.TINY_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43)
    unet, vae = UNet2DConditionModel(ucfg, usd, DEV), AutoencoderKL(vcfg, vsd, DEV)
    pipe = StableDiffusionImg2ImgPipeline(vae, unet, tokenizer=_Tok(), text_encoder=_Enc(768))
    h, w = z["image_hw"].tolist()
    img = Image.fromarray((torch.rand((h, w, 3), generator=torch.Generator().manual_seed(int(z["image_seed"]))) * 255).to(torch.uint8).numpy(), "RGB")
    cached = (torch.randn((1, 5, ucfg["cross_attention_dim"]), generator=torch.Generator().manual_seed(int(z["cached_seed"]))) * 0.5).to(DEV)
    src, dst, dst2 = str(tmp_path / "in.png"), str(tmp_path / "out.png"), str(tmp_path / "copy.png")
    img.save(src)
    U.copy_or_convert_image(img, src, dst, pipeline=pipe, unet=TextAlignedUNet(unet, cached), use_diffusion=True)
    out = np.asarray(Image.open(dst).convert("RGB"), np.uint8)
    assert list(out.shape) == z["out_shape"].tolist()
    dcrop = np.abs(out[480:544, 480:544].astype(int) - z["crop"].astype(int))
    pooled = torch.nn.functional.avg_pool2d(torch.from_numpy(out.copy()).permute(2, 0, 1).float()[None], 64)[0].numpy()
    print(f"V4 mirror vs the reference's PNG: crop max diff {dcrop.max()} (!=0: {(dcrop > 0).mean():.4f}); block means max diff {np.abs(pooled - z['pooled']).max():.3f} grey levels")
    assert dcrop.max() <= 1 and np.abs(pooled - z["pooled"]).max() <= 0.5
    U.copy_or_convert_image(img, src, dst2, pipeline=pipe, unet=unet, use_diffusion=False)
    assert open(src, "rb").read() == open(dst2, "rb").read()
    with pytest.raises(ValueError):   # without the wrapper the 1280-wide embeddings cannot reach a UNet with another cross_attention_dim
        U.copy_or_convert_image(img, src, dst, pipeline=pipe, unet=unet, use_diffusion=True)


def test_multimodal_augment_v7_against_oracle(tiny, monkeypatch):
    """Segmentor.ldiffusion_augment_for_multimodal (mirror of segmentor.py:301-386, sampler variant V7): RGB + depth -> posterior sample x
    0.18215 -> + Laplace(0, 1) x depth -> ControlNet residuals -> UNet -> z - eps x depth -> decode, against the same sequence on the oracle's
    graphs with the same random draws (the posterior's normal draw and the Laplace uniform draw are given: parity is defined given the draws)."""
    from ldiffusion_amd import models as M
    from ldiffusion_amd.segmentor import Segmentor
    g = torch.Generator().manual_seed(70)
    B = 2
    rgb, dtm = torch.rand((B, 3, 200, 180), generator=g), torch.rand((B, 1, 200, 180), generator=g)
    eps32 = torch.finfo(torch.float32).eps
    u = torch.rand((B, 4, 32, 32), generator=g) * (2 - eps32) + (eps32 - 1)
    post = torch.randn((B, 4, 32, 32), generator=g)
    ctx_dim = tiny["ucfg"]["cross_attention_dim"]
    shapes = tiny["unet"]._skip_shapes(1, 32, 32)

    def controlnet(sample, timestep, encoder_hidden_states, controlnet_cond, return_dict):   # the caller's module: a deterministic stand-in
        s = float(controlnet_cond.float().mean()) + float(sample.float().mean()) * 0.1
        gg = torch.Generator().manual_seed(5)
        return [(torch.randn(sh, generator=gg) * 0.2 * s).to(sample.device) for sh in shapes], (torch.randn(shapes[-1], generator=gg) * 0.2 * s).to(sample.device)

    calls = {"i": 0}
    def fake_sample(self, generator=None):
        i = calls["i"]; calls["i"] += 1
        return self.mean + self.std * post[i:i + 1].to(self.mean.device)
    monkeypatch.setattr(M._LatentDist, "sample", fake_sample)
    pipe = StableDiffusionImg2ImgPipeline(tiny["vae"], tiny["unet"], tokenizer=_Tok(), text_encoder=_Enc(48))
    seg = Segmentor(None, None, "cell", 3)
    torch.manual_seed(2)
    got = seg.ldiffusion_augment_for_multimodal(rgb, dtm, pipe, tiny["unet"], tiny["vae"], controlnet, B, DEV, u=u)
    proj = seg.ldiffusion_proj
    # the same on the oracle
    F_ = torch.nn.functional
    rgb2, dtm2 = F_.interpolate(rgb, size=(256, 256), mode="bilinear", align_corners=False), F_.interpolate(dtm, size=(256, 256), mode="bilinear", align_corners=False)
    emb = _Enc(48).table[torch.tensor([[49406, 320, 24857, 5471, 49407]])]
    ctx = F_.linear(emb, proj.weight.detach().cpu(), proj.bias.detach().cpu())
    assert ctx.shape[-1] == ctx_dim
    worst = 0.0
    for i in range(B):
        mom = tiny["opipe"].vae.encode(rgb2[i:i + 1]).latent_dist
        lat = (mom.mean + torch.exp(0.5 * torch.clamp(mom.logvar, -30.0, 20.0)) * post[i:i + 1]) * 0.18215
        depth = F_.interpolate(dtm2[i:i + 1], size=(32, 32), mode="bilinear", align_corners=False).repeat(1, 4, 1, 1)
        noisy = lat + noise_post.laplace_from_uniform(u[i:i + 1], 0.0, 1.0) * depth
        down, mid = controlnet(noisy, 1, ctx, dtm2[i:i + 1].repeat(1, 3, 1, 1), False)
        eps = tiny["opipe"].unet(noisy, 1, ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
        ref = tiny["opipe"].vae.decode((noisy - eps * depth) / 0.18215).sample[0].permute(1, 2, 0)
        e = rel_err(torch.from_numpy(got[i]), ref)
        worst = max(worst, e)
        assert got[i].shape == (256, 256, 3)
    print(f"V7 mirror (multimodal augment, ControlNet residuals): reconstruction rel err vs the oracle {worst:.3e}")
    assert worst <= 4e-3     # a decoded image: the tolerance of the VAE decode tests


def test_precision_modes_tiny(tiny):
    """ldiff_*_set_precision: 0 = all-fp16 storage (round-1 behaviour), 1 = split residual stream, 2 = every operand split.
    The error against the fp32 oracle must fall with the mode; the default (UNet 1, encoder 2, decoder 1) meets 1e-3."""
    g = torch.Generator().manual_seed(40)
    x = torch.randn((2, 4, 32, 32), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    img = torch.rand((2, 3, 128, 128), generator=g)
    z = torch.randn((2, 4, 16, 16), generator=g) * 0.5
    ref_u = tiny["opipe"].unet(x, 501, ctx).sample
    ref_e = tiny["opipe"].vae.encode(img).latent_dist.mean
    ref_d = tiny["opipe"].vae.decode(z).sample
    eu, ee, ed = [], [], []
    try:
        for mode in (0, 1, 2):
            tiny["unet"].set_precision(mode)
            tiny["vae"].set_precision(mode, mode)
            eu.append(rel_err(tiny["unet"](x.to(DEV), 501, ctx.to(DEV)).sample, ref_u))
            ee.append(rel_err(tiny["vae"].encode(img.to(DEV)).latent_dist.mean, ref_e))
            ed.append(rel_err(tiny["vae"].decode(z.to(DEV)).sample, ref_d))
    finally:
        tiny["unet"].set_precision(1)
        tiny["vae"].set_precision(2, 0)
    print(f"precision modes 0/1/2 (defaults: UNet 1, VAE encoder 2, decoder 0): unet {[f'{e:.2e}' for e in eu]}  vae encode {[f'{e:.2e}' for e in ee]}  vae decode {[f'{e:.2e}' for e in ed]}")
    assert eu[1] < eu[0] and eu[2] < eu[1] and ee[2] < ee[0] and ed[2] < ed[0]
    assert eu[1] <= 1e-3 and ee[2] <= 3e-4 and eu[2] <= 5e-4
    with pytest.raises(ValueError):
        tiny["unet"].set_precision(3)


def _assert_merged_mask_contract(mask, rmask, rmerged, W):
    """Merged-logit form of the mask contract: rmerged [C,H,W] float64 oracle logits after the Gaussian merge."""
    ys, xs = np.nonzero(mask != rmask)
    if len(ys) == 0:
        return
    c1, c2 = rmask[ys, xs].astype(int), mask[ys, xs].astype(int)
    margin = rmerged[c1, ys, xs] - rmerged[c2, ys, xs]
    bound = np.abs(W.double().numpy()[c1] - W.double().numpy()[c2]).sum(1)
    print(f"  mask contract (merged logits): {len(ys)} of {mask.size} pixels differ; worst (margin - bound) {(margin - bound).max():.2e}")
    assert (margin <= bound + 1e-5 * (np.abs(rmerged).max() + 1)).all()


def _probe_head(num_classes, n_feat, seed):
    """Deterministic linear probe [classes, passes] + bias on the per-pixel latent vectors (stand-in for the tissue head)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn((num_classes, n_feat), generator=g) / 255.0, torch.randn((num_classes,), generator=g) * 0.1


@pytest.mark.parametrize("step", [1.0, 0.5])
def test_config4_tiled_roi_20_passes_6_classes(tiny, step):
    """BASELINE.json configs[3] at reduced size (oracle in seconds): ROI -> tiles (nnU-Net origins) -> 20-pass sampler per tile
    -> 6-class logits per pixel -> Gaussian-weighted merge -> arg-max mask, against the CPU oracle run tile by tile."""
    from ldiffusion_amd import tiling
    from oracle import tiling as otiling
    N, C = 20, 6
    g = torch.Generator().manual_seed(4)
    roi = torch.rand((3, 128, 128), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    tiles, origins = tiling.split_tiles(roi.to(DEV), (64, 64), step)
    assert origins == otiling.tile_origins((128, 128), (64, 64), step) and tiles.shape[0] == (4 if step == 1.0 else 9)
    s = LaplaceSampler(tiny["pipe"])
    assert len(s.timesteps(N)) == N
    out = s.sample(tiles, ctx.to(DEV), N)
    W, bias = _probe_head(C, N, 5)
    feats = out["features"].float()                                                   # [n, N, 64, 64]
    logits = torch.einsum("cn,bnhw->bchw", W.to(DEV), feats) + bias.to(DEV)[None, :, None, None]
    merged = tiling.merge_tile_logits(logits, origins, (128, 128))
    mask = argmax_mask(merged[None])[0].cpu().numpy()
    # oracle, tile by tile
    ref = op.sample_v6(tiny["opipe"], tiles.cpu(), ctx, N)
    assert tiny["opipe"].unet.calls[-N:] == s.timesteps(N)
    e = rel_err(out["latents"], ref["latents"][-1])
    fd = np.abs(out["features"].cpu().numpy().astype(int) - ref["features"].astype(int))
    rlogits = np.einsum("cn,bnhw->bchw", W.numpy().astype(np.float64), ref["features"].astype(np.float64)) + bias.numpy()[None, :, None, None]
    rmerged = otiling.merge_logits(rlogits, origins, (128, 128))
    rmask = noise_post.argmax_mask(torch.from_numpy(rmerged)[None].float())[0]
    # the merge itself, on identical inputs, must agree to fp32 round-off
    m2 = otiling.merge_logits(logits.cpu().numpy().astype(np.float64), origins, (128, 128))
    assert np.abs(merged.cpu().numpy() - m2).max() <= 1e-4 * np.abs(m2).max()
    agree = (mask == np.asarray(rmask)).mean()
    print(f"config4 step={step}: {tiles.shape[0]} tiles x {N} passes; latents rel err {e:.3e}; luma max diff {fd.max()} (>1: {(fd > 1).mean():.4f}); "
          f"mask agreement {agree:.4f}")
    assert mask.shape == (128, 128) and mask.max() < C
    assert e <= 1e-3 and fd.max() <= 1
    # 20 passes of uint8 features: a luma off by one can move an arg-max, but only where the oracle's MERGED logits are within one grey level on
    # every feature of a tie (the Gaussian merge is a convex combination of tile logits, so the per-tile bound carries over)
    _assert_merged_mask_contract(mask, np.asarray(rmask), rmerged, W)
    assert agree >= 0.995, f"mask agreement {agree:.4f} with the oracle fell below the regression bound"
    if step == 1.0:   # non-overlapping: the merged mask is the tile masks side by side
        tm = argmax_mask(logits)
        assert torch.equal(tiling.merge_tile_masks(tm, origins, (128, 128)).cpu(), torch.from_numpy(mask))


@pytest.mark.timeout(2400)
def test_config3_full_size_roi_1024_four_tiles_20_passes():
    """BASELINE.json configs[3] at its real size: a 1024x1024 ROI tiled into 4 x 512x512 patches (nnU-Net origins at step 1.0), 20
    passes, SD-v1.5-width UNet + VAE, 6-class arg-max of a probe head over the per-pixel latent vectors.  The fp32 oracle runs tile 0
    through all 20 passes (~80 s on the box's cores); the other three tiles are covered by the batch-invariance of the sampler (a
    tile sampled alone vs inside the batch of four: no cross-sample coupling; the launches pick tile shapes / split-K plans by grid
    size, so the two runs round different fp16 operands: they agree to the same 1e-3 the oracle comparison asserts -- measured 2.5e-4
    after 20 passes -- and to one grey level), so the whole ROI is pinned by one oracle tile."""
    from ldiffusion_amd import tiling
    ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    N, C = 20, 6
    g = torch.Generator().manual_seed(44)
    roi = torch.rand((3, 1024, 1024), generator=g)
    ctx = torch.randn((1, 6, 768), generator=g) * 0.5
    pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, DEV), UNet2DConditionModel(ucfg, usd, DEV))
    s = LaplaceSampler(pipe)
    tiles, origins = tiling.split_tiles(roi.to(DEV), (512, 512), 1.0)
    assert origins == [(0, 0), (0, 512), (512, 0), (512, 512)] and len(s.timesteps(N)) == N
    out = s.sample(tiles, ctx.to(DEV), N)
    for k in (1, 3):                                            # batch invariance: tile k alone vs tile k inside the batch
        alone = s.sample(tiles[k:k + 1].contiguous(), ctx.to(DEV), N)
        dl = rel_err(alone["latents"][0], out["latents"][k])
        df = (alone["features"][0].int() - out["features"][k].int()).abs()
        print(f"  tile {k} alone vs in the batch: latents {dl:.2e}, luma max diff {int(df.max())} (!=0: {(df > 0).float().mean().item():.5f})")
        assert dl <= 1e-3 and int(df.max()) <= 1
    W, bias = _probe_head(C, N, 5)
    logits = torch.einsum("cn,bnhw->bchw", W.to(DEV), out["features"].float()) + bias.to(DEV)[None, :, None, None]
    mask = tiling.merge_tile_masks(argmax_mask(logits), origins, (1024, 1024)).cpu().numpy()
    merged = tiling.merge_tile_logits(logits, origins, (1024, 1024))
    assert np.array_equal(argmax_mask(merged[None])[0].cpu().numpy(), mask)       # non-overlapping tiles: the Gaussian merge changes nothing
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    opipe = op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg))
    ref = op.sample_v6(opipe, tiles[:1].cpu(), ctx, N)
    e = rel_err(out["latents"][:1], ref["latents"][-1])
    fd = np.abs(out["features"][:1].cpu().numpy().astype(int) - ref["features"].astype(int))
    rlogits = torch.einsum("cn,bnhw->bchw", W, torch.from_numpy(ref["features"]).float()) + bias[None, :, None, None]
    rmask0 = np.asarray(noise_post.argmax_mask(rlogits))[0]
    agree, ndiff = (mask[:512, :512] == rmask0).mean(), int((mask[:512, :512] != rmask0).sum())
    print(f"configs[3] full size: 4 tiles x {N} passes at SD15 width; tile 0 vs oracle: latents rel err {e:.3e}; luma max diff {fd.max()} "
          f"(!=0: {(fd > 0).mean():.4f}); mask agreement {agree:.5f} ({ndiff} of {rmask0.size} pixels differ)")
    print(f"  final latents of tile 0: {err_report(out['latents'][:1], ref['latents'][-1])}")
    # "Identical arg-max masks" is NOT reached on this config: the feature vector of a pixel is 20 uint8 luma values, each within ONE grey level of
    # the oracle's (asserted: any float error moves some rounding boundaries; 3-7 % of the values differ by one), and a linear head over 20 such
    # features changes its arg-max where two classes are within that margin (0.08-0.26 % of the pixels on this input, printed).  Asserted, per
    # differing pixel: the oracle's margin there is within what the pixel's own feature differences can move (the contract at the top of the file).
    assert mask.shape == (1024, 1024) and e <= 1e-3 and fd.max() <= 1
    assert assert_mask_flips_within_margin(mask[None, :512, :512], rmask0[None], ref["features"], W, bias, out["features"][:1].cpu().numpy(),
                                           "configs[3] tile 0") == ndiff
    assert agree >= MASK_AGREE_20_PASSES, f"mask agreement {agree:.5f} (measured 0.9992): regression of the off-by-one rate"


def test_tiles_are_independent_units(tiny):
    """Tiles shard like patches: no cross-sample coupling, so ranks can take disjoint tile ranges (parallel.shard_range) with no data-path collective.
    What holds bit for bit: the same sub-batch sampled twice, and a sub-batch against the same tiles at the same batch size elsewhere (equal shards, the
    bench's case).  What holds only to rounding: a tile alone against the same tile inside a LARGER batch -- the split-K plans (common.h splitk_by_model)
    and the small-batch GroupNorm path (kernels_norm.hip) are chosen by the launch's workgroup count, which depends on the batch, so fp32 partial sums
    associate differently (as they do in the reference: torch's kernels are not batch-invariant either), and a 1e-7 difference in a GroupNorm scale
    flips fp16 roundings downstream: the two runs end as far apart as either is from the fp32 oracle (measured: latents 3.2e-4 of range, 6.4 % of the
    luma values one grey level apart -- the same figures as against the oracle).  Asserted: the oracle contract itself, 1e-3 and one grey level."""
    from ldiffusion_amd import parallel, tiling
    g = torch.Generator().manual_seed(6)
    roi = torch.rand((3, 128, 128), generator=g).to(DEV)
    ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
    tiles, origins = tiling.split_tiles(roi, (64, 64), 1.0)
    s = LaplaceSampler(tiny["pipe"])
    o = s.sample(tiles, ctx, 5)
    full, full_lat = o["features"].clone(), o["latents"].clone()
    parts, lats = [], []
    for r in range(2):
        lo, hi = parallel.shard_range(tiles.shape[0], r, 2)
        o = s.sample(tiles[lo:hi].contiguous(), ctx, 5)
        parts.append(o["features"].clone()); lats.append(o["latents"].clone())
        again = s.sample(tiles[lo:hi].contiguous(), ctx, 5)
        assert torch.equal(again["features"], parts[-1]) and torch.equal(again["latents"], lats[-1])   # deterministic at a given batch size
    # equal shards see the same plans: shard 1's tiles sampled in shard 0's place (same batch size) give shard 1's results
    fd = (torch.cat(parts, 0).int() - full.int()).abs()
    e = rel_err(torch.cat(lats, 0), full_lat)
    print(f"tiles alone vs inside the batch of {tiles.shape[0]}: latents {e:.2e} of range, luma max diff {int(fd.max())} (!=0: {(fd > 0).float().mean().item():.5f})")
    assert int(fd.max()) <= 1 and (fd > 0).float().mean().item() <= 0.12 and e <= 1e-3


@pytest.mark.timeout(900)
def test_one_pass_1024_roi_against_oracle(tiny):
    """the reference-faithful single-pass form (segmentor.py:86-112, 490-545) on a full 1024x1024 ROI:
    128x128 latents, 16,384 tokens in the level-0 self-attention."""
    g = torch.Generator().manual_seed(8)
    x = torch.rand((1, 3, 1024, 1024), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    out = LaplaceSampler(tiny["pipe"]).sample(x.to(DEV), ctx.to(DEV), 1)
    ref = op.sample_one_pass(tiny["opipe"], x, ctx)
    e = rel_err(out["latents"], ref["latents"])
    rd = np.abs(out["rgb"].cpu().numpy().astype(int) - ref["rgb_u8"].astype(int))
    print(f"1024^2 one pass: latents rel err {e:.3e}; rgb max diff {rd.max()} (>1: {(rd > 1).mean():.5f})")
    assert out["rgb"].shape == (1, 1024, 1024, 3) or out["rgb"].shape[1:3] == (1024, 1024)
    assert e <= 1e-3 and rd.max() <= 1


def test_decode_side_stream_changes_nothing(tiny):
    """ldiff_sample with the VAE decodes on the side stream (default) == everything on the caller's stream, bit for bit, and
    back-to-back calls on one pipeline do not race on the shared VAE workspace."""
    g = torch.Generator().manual_seed(11)
    x = torch.rand((3, 3, 64, 64), generator=g).to(DEV)
    ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
    s = LaplaceSampler(tiny["pipe"])
    a1 = s.sample(x, ctx, 6)
    a2 = s.sample(x, ctx, 6)          # immediately again: the next encode must wait for the previous decodes
    s.set_overlap(0)
    b = s.sample(x, ctx, 6)
    s.set_overlap(1)
    torch.cuda.synchronize()
    for k in ("latents", "features", "rgb"):
        assert torch.equal(a1[k], b[k]) and torch.equal(a2[k], b[k]), k


def test_deferred_join_pipelines_batches(tiny):
    """Mode 2: two samplers on the same unet / vae used alternately, join deferred -- the encoder and UNet passes of batch k+1
    run under the trailing decodes of batch k.  Every batch must equal its serial result bit for bit; a standalone
    decode_latents in between (same decoder workspace) must not race either."""
    g = torch.Generator().manual_seed(12)
    batches = [torch.rand((2, 3, 64, 64), generator=g).to(DEV) for _ in range(5)]
    ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
    ref_s = LaplaceSampler(tiny["pipe"])
    ref_s.set_overlap(0)
    refs = [ref_s.sample(x, ctx, 5) for x in batches]
    zs = torch.randn((1, 4, 8, 8), generator=g).to(DEV)
    ref_img = tiny["pipe"].decode_latents(zs)
    torch.cuda.synchronize()
    ss = [LaplaceSampler(tiny["pipe"]), LaplaceSampler(tiny["pipe"])]
    for s in ss:
        s.set_overlap(2)
    outs, pending = [], None
    for k, x in enumerate(batches):
        cur = (ss[k % 2], ss[k % 2].sample(x, ctx, 5))
        if k == 2:
            img = tiny["pipe"].decode_latents(zs)     # standalone decode while sampler decodes are in flight
            assert np.array_equal(img, ref_img)
        if pending is not None:
            pending[0].join()
            outs.append({k2: v.clone() for k2, v in pending[1].items()})
        pending = cur
    pending[0].join()
    outs.append(pending[1])
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        for k2 in ("latents", "features", "rgb"):
            assert torch.equal(o[k2], r[k2]), k2
    with pytest.raises(RuntimeError):
        ss[0].sample(batches[0], ctx, 5)
        ss[0].set_overlap(1)                          # a join is pending: changing the mode is refused
    ss[0].join()
    ss[0].set_overlap(1)


def test_sampler_non_square_padded_prompt_per_sample_context(tiny):
    """Non-square patches (64 x 192), the padded 77-token prompt and one context per sample through the fused sampler."""
    g = torch.Generator().manual_seed(21)
    x = torch.rand((3, 3, 64, 192), generator=g)
    ctx = torch.randn((3, 77, 64), generator=g) * 0.5
    out = LaplaceSampler(tiny["pipe"]).sample(x.to(DEV), ctx.to(DEV), 4)
    ref = op.sample_v6(tiny["opipe"], x, ctx, 4)
    e = rel_err(out["latents"], ref["latents"][-1])
    fd = np.abs(out["features"].cpu().numpy().astype(int) - ref["features"].astype(int))
    print(f"sampler 64x192, L=77, per-sample ctx: latents rel err {e:.3e}; luma max diff {fd.max()} (>1: {(fd > 1).mean():.4f})")
    assert out["features"].shape == (3, 4, 64, 192) and e <= 1e-3 and fd.max() <= 1


def test_bilinear_resize_matches_torch():
    from ldiffusion_amd.pipeline import bilinear_resize
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(31)
    for shape, size in [((2, 3, 512, 512), 64), ((1, 3, 64, 64), (64, 64)), ((2, 3, 40, 56), (64, 64)), ((1, 1, 7, 5), (3, 9)), ((1, 3, 64, 64), (1024, 1024))]:
        x = torch.randn(shape, generator=g)
        got = bilinear_resize(x.to(DEV), size).cpu()
        ref = F.interpolate(x, size=size if isinstance(size, tuple) else (size, size), mode="bilinear", align_corners=False)
        assert got.shape == ref.shape and (got - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item()), shape
    with pytest.raises(ValueError):
        bilinear_resize(torch.zeros((3, 4, 4), device=DEV), 2)


def test_training_time_features_v5_against_oracle(tiny):
    """SURVEY F9/F10 (ldiffusion.py:228-247): Laplace forward noise on the fixed z0 -> UNet -> decode of the UNet output ->
    bilinear 64 x 64 -> float luma, concatenated over the scheduler timesteps; parity given the uniform draws."""
    from ldiffusion_amd.pipeline import laplace_features
    g = torch.Generator().manual_seed(32)
    x = torch.rand((2, 3, 128, 128), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    n_sched = 4
    tiny["opipe"].scheduler.set_timesteps(n_sched)
    nts = len(tiny["opipe"].scheduler.timesteps)
    eps32 = torch.finfo(torch.float32).eps
    u_list = [torch.rand((2, 4, 16, 16), generator=g) * (2 - eps32) + (eps32 - 1) for _ in range(nts)]
    got = laplace_features(tiny["pipe"], x.to(DEV), ctx.to(DEV), n_sched, u_list=[u.to(DEV) for u in u_list], out_hw=64)
    ref = op.laplace_features_v5(tiny["opipe"], x, ctx, n_sched, u_list, out_hw=64)
    eg, er = rel_err(got["gray"], ref["gray"]), rel_err(got["rgb"], ref["rgb"])
    print(f"V5 features: {nts} planes, gray rel err {eg:.3e}, last rgb rel err {er:.3e}")
    # the decoded image of the raw UNet output (no 1/scaling_factor, ldiffusion.py:240) through the decoder: image-range error
    # (measured 2.5e-3 / 3.1e-3 of the range)
    assert got["gray"].shape == (2, nts, 64, 64) and eg <= 5e-3 and er <= 5e-3


def _fixture_context(z):
    """The text conditioning of the reference fixture: token table lookup + the 768->cad projection (segmentor.py:54-60)."""
    table = torch.randn((49408, int(z["hidden"])), generator=torch.Generator().manual_seed(99)) * 0.5
    emb = table[torch.tensor([z["ids"].tolist()])]
    return torch.nn.functional.linear(emb, torch.from_numpy(z["proj_weight"]), torch.from_numpy(z["proj_bias"]))


def test_segmentor_mirror_augment_against_reference_fixture(tiny):
    """ldiffusion_amd.Segmentor.ldiffusion_augment (batched, on the device) against tests/golden/reference_augment_v3.npz, which
    was produced by the REFERENCE's own Segmentor.ldiffusion_augment (segmentor.py:86-112) driven with the oracle objects."""
    import os
    from ldiffusion_amd.segmentor import Segmentor, TextAlignedUNet
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_augment_v3.npz"))
    inputs = torch.rand((2, 3, 64, 64), generator=torch.Generator().manual_seed(int(z["inputs_seed"])))
    ctx = _fixture_context(z).to(DEV)
    seg = Segmentor(None, None, "cell", 3)
    out = seg.ldiffusion_augment(inputs, tiny["pipe"], tiny["unet"], tiny["vae"], text_embeddings=ctx)
    assert list(out.shape) == z["out_shape"].tolist() and out.dtype == torch.float32 and out.is_cuda
    pooled = torch.nn.functional.avg_pool2d(out, 64).cpu().numpy()
    d = np.abs(pooled - z["pooled"]).max()
    print(f"segmentor mirror vs reference fixture: max |pooled diff| = {d:.2e}")
    assert d < 4e-3                                            # mean of a 64x64 block of uint8/255 pixels, each <= 1 grey level off
    # F12: the text-align wrapper substitutes its cached embeddings when the caller passes None or a wrong width
    wrapped = TextAlignedUNet(tiny["unet"], ctx)
    x = torch.randn((2, 4, 8, 8), generator=torch.Generator().manual_seed(1)).to(DEV)
    ref = tiny["unet"](x, 1, ctx).sample
    assert torch.equal(wrapped(x, 1, None).sample, ref)
    assert torch.equal(wrapped(x, 1, torch.zeros((2, 5, 7), device=DEV)).sample, ref)
    with pytest.raises(ValueError):
        seg.initialize_model("organ", 3)
    with pytest.raises(RuntimeError):
        seg.inference_cell_model("x.png", "sd", "w", None)     # no head: refuses instead of substituting one


def test_segmentor_mirror_build_augmented_dataloader(tiny):
    """segmentor.py:144-161 through the mirror: the augmented inputs of every batch (here: `ldiffusion_augment` itself, the reference's
    caller passes it as `augment_fn`) and the masks, cached on the host and served by a shuffling loader of the requested batch size."""
    from functools import partial
    from ldiffusion_amd.segmentor import Segmentor
    g = torch.Generator().manual_seed(12)
    ctx = (torch.randn((1, 6, 64), generator=g) * 0.5).to(DEV)
    batches = [(torch.rand((n, 3, 64, 64), generator=g), torch.randint(0, 3, (n, 64, 64), generator=g), None) for n in (2, 1, 2)]
    seg = Segmentor(None, None, "cell", 3)
    aug = partial(seg.ldiffusion_augment, text_embeddings=ctx)
    torch.manual_seed(0)
    loader = seg.build_augmented_dataloader(batches, aug, tiny["pipe"], tiny["unet"], tiny["vae"], DEV, batch_size=2, category="train")
    got = list(loader)
    assert [len(b[0]) for b in got] == [2, 2, 1] and all(not b[0].is_cuda and b[0].shape[1:] == (3, 1024, 1024) for b in got)
    want_x = torch.cat([aug(x, tiny["pipe"], tiny["unet"], tiny["vae"]).cpu() for x, _, _ in batches])
    want_m = torch.cat([m for _, m, _ in batches])
    xs, ms = torch.cat([b[0] for b in got]), torch.cat([b[1] for b in got])
    key = lambda m: tuple(m.reshape(-1)[:64].tolist())            # the masks identify the samples whatever the shuffle did
    order = [[key(m) for m in want_m].index(key(m)) for m in ms]
    assert sorted(order) == list(range(5))
    assert torch.equal(xs, want_x[order]) and torch.equal(ms, want_m[order])


def test_segmentor_mirror_inference_cell_model_end_to_end(tiny, tmp_path):
    """segmentor.py:490-545 through the mirror: checkpoint directories in the diffusers layout -> load_ldiffusion -> 1024x1024
    single pass -> decoded RGB handed to the head ON THE DEVICE -> arg-max -> NEAREST resize to the input size."""
    import json
    from PIL import Image
    from ldiffusion_amd.segmentor import IMAGENET_MEAN, IMAGENET_STD, Segmentor
    sd_dir, w_dir = tmp_path / "sd", tmp_path / "train_save" / "unet" / "25_01_01"
    tiny["unet"].save_pretrained(str(sd_dir / "unet"))
    tiny["vae"].save_pretrained(str(sd_dir / "vae"))
    tiny["unet"].save_pretrained(str(w_dir))
    rng = np.random.default_rng(5)
    img = (rng.random((96, 80, 3)) * 255).astype(np.uint8)
    path = tmp_path / "roi.png"
    Image.fromarray(img).save(path)
    ctx = (torch.randn((1, 6, 64), generator=torch.Generator().manual_seed(2)) * 0.5).to(DEV)
    seen = {}

    def head(x):                                               # stand-in for CellSegClassifier: 3-class logits from the decoded image
        seen["device"], seen["shape"] = x.device.type, tuple(x.shape)
        return torch.stack([x[:, 0], x[:, 1] * 0.5, -x[:, 2]], 1)

    seg = Segmentor(None, None, "cell", 3)
    decoded, mask = seg.inference_cell_model(str(path), str(sd_dir), str(w_dir), None, head=head, text_embeddings=ctx)
    assert seen == {"device": "cuda", "shape": (1, 3, 1024, 1024)}
    assert decoded.size == (80, 96) and mask.shape == (96, 80) and mask.dtype == np.uint8 and mask.max() <= 2
    # the same path by hand on the oracle
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    x = torch.from_numpy(np.asarray(Image.fromarray(img).resize((1024, 1024), Image.BILINEAR), np.float32) / 255.0).permute(2, 0, 1)[None]
    r = op.sample_one_pass(tiny["opipe"], (x - mean) / std, ctx.cpu())
    logits = head((torch.from_numpy(r["rgb_u8"]).permute(0, 3, 1, 2).float() / 255.0 - mean) / std)
    rmask = np.array(Image.fromarray(np.asarray(noise_post.argmax_mask(logits)[0]).astype(np.uint8)).resize((80, 96), resample=Image.NEAREST))
    agree = (mask == rmask).mean()
    print(f"inference_cell_model mirror: mask agreement with the oracle {agree:.4f}")
    assert agree > 0.995


def _tiny_clip_dirs(root, hidden=32):
    """A real (tiny) CLIP tokenizer + CLIPTextModel written to <root>/tokenizer and <root>/text_encoder with `transformers`, so that
    the reference's text path (tokenizer -> text_encoder -> proj, segmentor.py:31-60) runs end to end offline."""
    import json
    import os
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    chars = list("abcdefghijklmnopqrstuvwxyz")
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": len(chars) + i for i, c in enumerate(chars)})
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = len(vocab), len(vocab) + 1
    td = os.path.join(root, "tokenizer")
    os.makedirs(td)
    json.dump(vocab, open(os.path.join(td, "vocab.json"), "w"))
    open(os.path.join(td, "merges.txt"), "w").write("#version: 0.2\n")
    CLIPTokenizer(os.path.join(td, "vocab.json"), os.path.join(td, "merges.txt")).save_pretrained(td)
    torch.manual_seed(0)
    cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=2, num_attention_heads=4,
                         max_position_embeddings=77, bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"],
                         pad_token_id=vocab["<|endoftext|>"])
    CLIPTextModel(cfg).save_pretrained(os.path.join(root, "text_encoder"))


def _write_sd_dirs(tmp_path, tiny, unet_dtype=torch.float16, vae_dtype=torch.bfloat16):
    """SD directory + fine-tuned UNet directory in the diffusers layout the reference reads (ldiffusion.py:67,273-277; segmentor.py:77-80),
    with config.json files TRIMMED of every field that has a diffusers default, fp16 / bf16 safetensors, and proj_weights.pt."""
    import json
    from safetensors.torch import save_file
    sd_dir, w_dir = tmp_path / "sd", tmp_path / "train_save" / "unet" / "25_01_01"
    for d, cfg, sd, dt in ((sd_dir / "unet", tiny["ucfg"], tiny["usd"], unet_dtype), (sd_dir / "vae", tiny["vcfg"], tiny["vsd"], vae_dtype),
                           (w_dir, tiny["ucfg"], tiny["usd"], unet_dtype)):
        d.mkdir(parents=True)
        keep = ("_class_name", "block_out_channels", "cross_attention_dim", "down_block_types", "up_block_types", "layers_per_block", "latent_channels")
        json.dump({k: v for k, v in cfg.items() if k in keep}, open(d / "config.json", "w"))   # no scaling_factor, norm_eps, flip_sin_to_cos, ...
        save_file({k: v.to(dt).contiguous() for k, v in sd.items()}, str(d / weights.WEIGHTS_NAME))
    _tiny_clip_dirs(str(sd_dir))
    g = torch.Generator().manual_seed(77)
    proj = {"weight": torch.randn((64, 32), generator=g) * 0.2, "bias": torch.randn((64,), generator=g) * 0.1}
    torch.save(proj, str(w_dir / "proj_weights.pt"))
    return sd_dir, w_dir, proj


def test_text_path_and_trimmed_configs_end_to_end(tiny, tmp_path):
    """(a) `_get_text_embeddings` executed for real: tokenizer (unpadded) -> CLIP text encoder -> Linear(768->cad)
    loaded strictly from proj_weights.pt (segmentor.py:31-60); (b) a vae/config.json without `scaling_factor` and a unet/config.json
    without the defaulted fields load with diffusers' defaults, from fp16 and bf16 safetensors."""
    from transformers import CLIPTextModel, CLIPTokenizer
    from ldiffusion_amd.segmentor import Segmentor
    sd_dir, w_dir, proj = _write_sd_dirs(tmp_path, tiny)
    seg = Segmentor(None, None, "cell", 3)
    pipeline, unet, vae = seg.load_ldiffusion(str(w_dir), str(sd_dir))
    assert vae.config.scaling_factor == 0.18215 and unet.config.norm_eps == 1e-5 and unet.config.flip_sin_to_cos is True
    assert unet.config.cross_attention_dim == 64 and pipeline.text_encoder.config.hidden_size == 32
    emb = seg._get_text_embeddings("A pathological slide", 2, pipeline, unet)
    tok = CLIPTokenizer.from_pretrained(str(sd_dir / "tokenizer"))
    enc = CLIPTextModel.from_pretrained(str(sd_dir / "text_encoder"))
    ids = torch.tensor(tok(["A pathological slide"] * 2)["input_ids"])
    assert ids.shape[1] == 20                                    # unpadded: <bos> + 18 characters + <eos>
    with torch.no_grad():
        ref = torch.nn.functional.linear(enc(ids)["last_hidden_state"], proj["weight"], proj["bias"])
    assert emb.shape == (2, 20, 64) and emb.is_cuda and (emb.cpu() - ref).abs().max() <= 1e-4 * ref.abs().max()
    # the loaded projection is the saved one (strict): a checkpoint with a foreign key is refused
    assert torch.equal(seg.ldiffusion_proj.weight.cpu(), proj["weight"])
    torch.save({"weight": proj["weight"], "bias": proj["bias"], "extra": torch.zeros(1)}, str(w_dir / "proj_weights.pt"))
    with pytest.raises(RuntimeError):
        Segmentor(None, None, "cell", 3).load_ldiffusion(str(w_dir), str(sd_dir))
    torch.save(proj, str(w_dir / "proj_weights.pt"))
    # the checkpoints were written in fp16 / bf16: the loaded graphs equal graphs built from those rounded values
    x = torch.randn((1, 4, 16, 16), generator=torch.Generator().manual_seed(3)).to(DEV)
    u16 = UNet2DConditionModel(tiny["ucfg"], {k: v.to(torch.float16).float() for k, v in tiny["usd"].items()}, DEV)
    assert torch.equal(unet(x, 501, emb[:1]).sample, u16(x, 501, emb[:1]).sample)
    vbf = AutoencoderKL(tiny["vcfg"], {k: v.to(torch.bfloat16).float() for k, v in tiny["vsd"].items()}, DEV)
    z = torch.randn((1, 4, 8, 8), generator=torch.Generator().manual_seed(4)).to(DEV)
    assert torch.equal(vae.decode(z).sample, vbf.decode(z).sample)
    # decode_latents reads vae.config.scaling_factor (the field the trimmed config.json lacks)
    assert np.array_equal(pipeline.decode_latents(z), StableDiffusionImg2ImgPipeline(vbf, u16).decode_latents(z))


def test_tissue_inference_mirror(tiny, tmp_path):
    """segmentor.py:388-488 through the mirror and through LDiffusionModel.inference(level="tissue"): a square image goes through the
    one-pass sampler, a non-square one skips the diffusion (:427,449-450), a folder returns (None, None) (:421); the decoded RGB is
    handed to the injected tissue head ON THE DEVICE through the sliding-window predictor (Gaussian fp16 accumulation, mirroring)."""
    from PIL import Image
    from ldiffusion_amd import tiling
    from ldiffusion_amd.ldiffusion import LDiffusionModel
    from ldiffusion_amd.segmentor import IMAGENET_MEAN, IMAGENET_STD, Segmentor
    sd_dir, w_dir, proj = _write_sd_dirs(tmp_path, tiny, torch.float32, torch.float32)
    rng = np.random.default_rng(7)
    sq, rect = tmp_path / "sq.png", tmp_path / "rect.png"
    Image.fromarray((rng.random((96, 96, 3)) * 255).astype(np.uint8)).save(sq)
    Image.fromarray((rng.random((80, 120, 3)) * 255).astype(np.uint8)).save(rect)
    Wh = torch.tensor([[1., -1., 0.], [0., 1., -1.], [-1., 0., 1.], [0.5, 0.5, -1.]])
    seen = []

    def predictor(x):                                            # stand-in tissue head: 4-class logits, not mirror-equivariant
        seen.append((x.device.type, tuple(x.shape)))
        ramp = torch.arange(x.shape[-1], dtype=torch.float32, device=x.device) * 0.01
        return torch.einsum("oc,bchw->bohw", Wh.to(x.device), x.float() / 255.0) + ramp[None, None, None, :]

    model = LDiffusionModel(str(sd_dir), "tissue")
    decoded, mask = model.inference(str(sq), str(w_dir), None, 4, predictor=predictor)
    assert decoded.size == (1024, 1024) and mask.shape == (1024, 1024) and mask.dtype == np.uint8 and mask.max() <= 3
    assert seen[0] == ("cuda", (1, 3, 512, 512)) and len(seen) == 9 * 4   # 3x3 tiles at step 0.5, x4 mirror combinations
    # the same by hand on the oracle: text path on the CPU, one-pass sampler, the same head through the same tiling code on the CPU
    from transformers import CLIPTextModel, CLIPTokenizer
    tok, enc = CLIPTokenizer.from_pretrained(str(sd_dir / "tokenizer")), CLIPTextModel.from_pretrained(str(sd_dir / "text_encoder"))
    with torch.no_grad():
        ctx = torch.nn.functional.linear(enc(torch.tensor(tok(["A pathological slide"])["input_ids"]))["last_hidden_state"], proj["weight"], proj["bias"])
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    x = torch.from_numpy(np.asarray(Image.open(sq).convert("RGB").resize((1024, 1024), Image.BILINEAR), np.float32) / 255.0).permute(2, 0, 1)[None]
    r = op.sample_one_pass(tiny["opipe"], (x - mean) / std, ctx)
    rgbd = np.abs(np.asarray(decoded).astype(int) - r["rgb_u8"][0].astype(int))
    rl = tiling.predict_sliding_window_return_logits(torch.from_numpy(r["rgb_u8"][0]).permute(2, 0, 1).float(), predictor, 4, (512, 512), 0.5, True, (0, 1))
    rmask = np.asarray(noise_post.argmax_mask(rl[None].float())[0])
    agree = (mask == rmask).mean()
    print(f"tissue mirror: decoded rgb max diff {rgbd.max()}, mask agreement with the oracle {agree:.4f}")
    assert rgbd.max() <= 1 and agree > 0.995
    # non-square: the diffusion is skipped, the image itself goes to the head -> exactly the CPU result
    seg = Segmentor(None, None, "tissue", 4)
    dec2, mask2 = seg.inference_tissue_model_nnUNetv2(str(rect), str(sd_dir), str(w_dir), None, predictor=predictor, tile_size=(64, 64))
    img = torch.from_numpy(np.asarray(Image.open(rect).convert("RGB"), np.uint8)).permute(2, 0, 1).float()
    l2 = tiling.predict_sliding_window_return_logits(img, predictor, 4, (64, 64), 0.5, True, (0, 1))
    assert dec2.size == (120, 80) and np.array_equal(mask2, np.asarray(noise_post.argmax_mask(l2[None].float())[0]))
    # folder mode
    folder, outdir = tmp_path / "imgs", tmp_path / "pred"
    folder.mkdir()
    Image.open(rect).save(folder / "case_0000.png")
    with pytest.raises(ValueError, match="output_path must be specified"):
        seg.inference_tissue_model_nnUNetv2(str(folder), str(sd_dir), str(w_dir), None, predictor=predictor)
    assert seg.inference_tissue_model_nnUNetv2(str(folder), str(sd_dir), str(w_dir), None, output_path=str(outdir), predictor=predictor,
                                               tile_size=(64, 64)) == (None, None)
    assert np.array_equal(np.asarray(Image.open(outdir / "case_0000.png")), mask2)
    with pytest.raises(RuntimeError):
        seg.inference_tissue_model_nnUNetv2(str(sq), str(sd_dir), str(w_dir), None)          # no head: refuses instead of substituting one


def test_sliding_window_predictor_on_device_matches_reference_fixture():
    """The device run of tiling.predict_sliding_window_return_logits against the REFERENCE's own predictor code
    (tests/golden/reference_sliding_window.npz; the CPU run of the same function is bit exact, tests/test_cpu_oracle.py): fp16
    accumulation in the same order.  torch's device kernels for half division / mixed-dtype multiply may differ from the CPU ones by
    one fp16 ulp on isolated elements, so: at most 1 ulp anywhere, and at most 0.1 % of the elements differ at all."""
    import os
    from ldiffusion_amd import tiling
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_sliding_window.npz"))
    W = torch.from_numpy(z["head_weight"]).to(DEV)

    def network(x):
        ramp = torch.arange(x.shape[-1], dtype=torch.float32, device=x.device) * 0.125
        return torch.einsum("oc,bchw->bohw", W, x.float()) + ramp[None, None, None, :]

    for tag in "abc":
        th, tw, step, mm = z[tag + "_cfg"]
        mirror = None if mm < 0 else tuple(i for i in range(2) if (int(mm) >> i) & 1)
        got = tiling.predict_sliding_window_return_logits(torch.from_numpy(z[tag + "_image"]).to(DEV), network, 4, (int(th), int(tw)), float(step), True, mirror)
        ref = torch.from_numpy(z[tag + "_logits_f16"])
        d = (got.cpu().float() - ref.float()).abs()
        ulp = torch.maximum(ref.float().abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
        print(f"sliding window {tag}: {int((d > 0).sum())} of {d.numel()} elements differ from the reference fixture")
        assert got.is_cuda and got.dtype == torch.float16 and bool((d <= ulp).all()) and (d > 0).float().mean() <= 1e-3, tag


def test_orchestrator_mirror_errors(tmp_path):
    """ldiffusion_amd.LDiffusionModel keeps the reference's constructor / inference signature and its error for a bad level
    (ldiffusion.py:32,317-324); what is outside the hot path raises instead of silently doing something else."""
    from ldiffusion_amd.ldiffusion import LDiffusionModel
    m = LDiffusionModel(str(tmp_path), "organ")
    with pytest.raises(ValueError, match="Invalid level specified"):
        m.inference("x.png", "w", None, 3, head=lambda x: x)
    with pytest.raises(RuntimeError):
        LDiffusionModel(str(tmp_path), "tissue").inference("x.png", "w", None, 6)      # no predictor given
    with pytest.raises(RuntimeError, match="train_loader"):
        LDiffusionModel(str(tmp_path), "cell", local_rank=0).train(None)                 # the dataset pipeline is injected, not rebuilt
    with pytest.raises(NotImplementedError):
        LDiffusionModel(str(tmp_path), "cell", local_rank=0).train(None, component="segmentor", ldiffusion_weight="w")


def test_train_ldiffusion_mirror_end_to_end(tiny, tmp_path):
    """LDiffusionModel.train(component="ldiffusion") (ldiffusion.py:121-295,297-305) with an injected loader: two epochs over two batches on
    the HIP kernels (forward and backward), CSV log, best-epoch checkpoint in the layout Segmentor.load_ldiffusion reads back."""
    import csv
    from types import SimpleNamespace
    from ldiffusion_amd.ldiffusion import LDiffusionModel
    from ldiffusion_amd.segmentor import Segmentor
    sd_dir, _, _ = _write_sd_dirs(tmp_path, tiny, torch.float32, torch.float32)
    g = torch.Generator().manual_seed(21)
    loader = [(torch.rand((2, 3, 96, 96), generator=g), None, torch.randint(0, 3, (2, 1, 96, 96), generator=g).to(torch.uint8)) for _ in range(2)]
    args = SimpleNamespace(diffusion_path=str(sd_dir), num_inference_steps=5, batch_size=2, ldiffusion_epochs=2, output_root=str(tmp_path / "out"))
    torch.manual_seed(5)
    model = LDiffusionModel(str(sd_dir), "cell")
    saved = model.train(args, component="ldiffusion", train_loader=loader)
    assert saved.startswith(str(tmp_path / "out")) and (tmp_path / "out" / "LDiffusion" / "train_save" / "unet").is_dir()
    import os
    assert {"config.json", weights.WEIGHTS_NAME, "proj_weights.pt"} <= set(os.listdir(saved))
    rows = list(csv.reader(open(next((tmp_path / "out" / "train_save" / "loss").glob("*/contrast_loss.csv")))))
    assert rows[0] == ["epoch", "loss"] and [r[0] for r in rows[1:]] == ["1", "2"] and all(np.isfinite(float(r[1])) for r in rows[1:])
    print(f"train_ldiffusion mirror: epoch losses {[round(float(r[1]), 4) for r in rows[1:]]}")
    # the checkpoint loads through the inference path and differs from the starting weights
    pipeline, unet, _ = Segmentor(None, None, "cell", 3).load_ldiffusion(saved, str(sd_dir))
    x = torch.randn((1, 4, 8, 8), generator=g).to(DEV)
    ctx = torch.randn((1, 6, 64), generator=g).to(DEV)
    assert not torch.equal(unet(x, 1, ctx).sample, tiny["unet"](x, 1, ctx).sample)
    assert tuple(torch.load(os.path.join(saved, "proj_weights.pt"))["weight"].shape) == (64, 32)
    # the same loop with the step driven from the Python tape (use_graph=False) instead of the captured graph: same draws, same losses
    args2 = SimpleNamespace(diffusion_path=str(sd_dir), num_inference_steps=5, batch_size=2, ldiffusion_epochs=2, output_root=str(tmp_path / "out2"), use_graph=False)
    torch.manual_seed(5)
    LDiffusionModel(str(sd_dir), "cell").train(args2, component="ldiffusion", train_loader=loader)
    rows2 = list(csv.reader(open(next((tmp_path / "out2" / "train_save" / "loss").glob("*/contrast_loss.csv")))))
    print(f"  eager loop: epoch losses {[round(float(r[1]), 4) for r in rows2[1:]]}")
    for a, b in zip(rows[1:], rows2[1:]):
        assert abs(float(a[1]) - float(b[1])) <= 2e-3 * abs(float(b[1])), (a, b)


@pytest.mark.parametrize("prec", [0, 2])
def test_decode_tail_fused_into_conv_out(tiny, prec):
    """decode_latents' tail ((x/2+0.5).clamp(0,1) -> (.*255).round() half-even -> PIL's integer luma) runs inside the epilogue of
    the VAE's conv_out (narrow-output kernel), from its fp32 sums.  With every output requested the fp32 tensor is written too: the image,
    uint8 and luma outputs must be, bit for bit, what the tensor formulation makes of that tensor; and the call that asks for the uint8
    outputs only (no fp32 tensor at all: the sampler's feature path) must return the same bytes."""
    vae = tiny["vae"]
    vae.set_precision(decoder=prec)
    try:
        g = torch.Generator().manual_seed(4 + prec)
        z = (torch.randn((3, 4, 16, 24), generator=g) * 1.5).to(DEV)
        luma = torch.full((3, 2, 128, 192), 7, dtype=torch.uint8, device=DEV)
        sample, image, rgb = vae._decode(z, 1.0 / 0.18215, want_sample=True, want_image=True, want_rgb=True, luma=luma, slot=1)
        x = sample.permute(0, 2, 3, 1)
        want_img = (x * 0.5 + 0.5).clamp(0, 1)
        want_rgb = (want_img * 255.0).round().to(torch.uint8)
        q = want_rgb.to(torch.int64)
        want_luma = ((19595 * q[..., 0] + 38470 * q[..., 1] + 7471 * q[..., 2] + 0x8000) >> 16).to(torch.uint8)
        assert torch.equal(image, want_img) and torch.equal(rgb, want_rgb)
        assert torch.equal(luma[:, 1], want_luma) and bool((luma[:, 0] == 7).all())
        luma2 = torch.zeros_like(luma)
        _, _, rgb2 = vae._decode(z, 1.0 / 0.18215, want_rgb=True, luma=luma2, slot=0)
        assert torch.equal(rgb2, rgb) and torch.equal(luma2[:, 0], want_luma)
    finally:
        vae.set_precision(decoder=0)


def test_probe_argmax_one_launch_bit_exact():
    """ldiff_probe_argmax_u8 (linear probe over the uint8 per-pixel latent vectors + arg-max in one launch, the bench's mask tail) against
    the host statement of the same pinned arithmetic (oracle.noise_post.probe_argmax), bit for bit, ties included."""
    from ldiffusion_amd.pipeline import probe_argmax_mask
    g = torch.Generator().manual_seed(77)
    for B, N, C, H, W in ((2, 5, 6, 32, 36), (1, 20, 11, 8, 4), (3, 1, 1, 16, 16), (1, 64, 32, 4, 4)):
        f = torch.randint(0, 256, (B, N, H, W), generator=g, dtype=torch.uint8)
        f[0, :, 0, :4] = 0                                     # all-zero vectors: the bias decides
        w = torch.randn((C, N), generator=g) / N ** 0.5
        b = torch.randn((C,), generator=g) * 0.1
        if C > 2:
            w[2] = w[1]; b[2] = b[1]                           # exact ties between classes 1 and 2 -> the lower index wins
        for bias in (b, None):
            got = probe_argmax_mask(f.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV)).cpu().numpy()
            ref = noise_post.probe_argmax(f.numpy(), w.numpy(), None if bias is None else bias.numpy())
            assert got.dtype == np.uint8 and np.array_equal(got, ref), (B, N, C, H, W)
            assert C <= 2 or not (got == 2).any()
    # the float formulation the bench used before (einsum + argmax) agrees except where two logits are within float round-off
    f = torch.randint(0, 256, (2, 5, 64, 64), generator=g, dtype=torch.uint8)
    w, b = torch.randn((6, 5), generator=g), torch.randn((6,), generator=g)
    lg = torch.einsum("cn,bnhw->bchw", w, f.float() * (1.0 / 255.0)) + b[None, :, None, None]
    assert (probe_argmax_mask(f.to(DEV), w.to(DEV), b.to(DEV)).cpu().numpy() != noise_post.argmax_mask(lg)).mean() <= 1e-3
    assert probe_argmax_mask(torch.zeros((0, 5, 4, 4), dtype=torch.uint8, device=DEV), w.to(DEV), b.to(DEV)).shape == (0, 4, 4)
    with pytest.raises(ValueError):
        probe_argmax_mask(f.to(DEV), torch.zeros((6, 4), device=DEV))
    with pytest.raises(ValueError):
        probe_argmax_mask(f.to(DEV), torch.zeros((40, 5), device=DEV))


@pytest.mark.timeout(2400)
def test_config1_b8_bench_mode_against_oracle():
    """The configuration bench.py TIMES (BASELINE.json configs[1]): SD-v1.5 width, 512x512, 5 passes, **B = 8**, two samplers with deferred
    joins alternating over the same unet / vae (two batches in flight, ldiff_pipeline_set_overlap 2: bench.py run_steps), the persistent conv
    kernels on short runs beside the UNet stream.  At B = 8 the kernel selection differs from the B = 2 test above (dataflow conv on the 64x64
    maps, other split-K plans).  Checked: every pipelined batch equals the serial (one stream) result bit for bit; patches 0 and 7 against the
    fp32 CPU oracle (latents <= 1e-3 of range, luma within one grey level, probe-head masks)."""
    from ldiffusion_amd.pipeline import probe_argmax_mask
    ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    B, N = 8, 5
    x = torch.rand((B, 3, 512, 512), generator=torch.Generator().manual_seed(1234))                 # bench.py's images / context
    ctx = torch.randn((1, 6, 768), generator=torch.Generator().manual_seed(1235)) * 0.5
    pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, DEV), UNet2DConditionModel(ucfg, usd, DEV))
    xd, cd = x.to(DEV), ctx.to(DEV)
    s1, s2 = LaplaceSampler(pipe), LaplaceSampler(pipe)
    s1.set_overlap(0)
    serial = s1.sample(xd, cd, N)
    serial = {k: v.clone() for k, v in serial.items()}
    s1.sample(xd, cd, N)                                                                                # second use: the UNet graph is captured / replayed from here on
    s1.set_overlap(2); s2.set_overlap(2)
    outs, prev = [], None
    for i in range(4):                                                                                  # bench.py run_steps
        sp = (s1, s2)[i & 1]
        out = sp.sample(xd, cd, N)
        if prev is not None:
            prev[0].join()
            outs.append({k: v.clone() for k, v in prev[1].items()})
        prev = (sp, out)
    prev[0].join()
    outs.append(prev[1])
    torch.cuda.synchronize()
    s1.set_overlap(1); s2.set_overlap(1)
    assert pipe.unet.graph_replays >= 5
    # launches of one UNet pass = nodes of the captured graph (round 6: 384 at this configuration; the round-5 verdict's bound is 550): a change that
    # splits a fused launch again, or sends the split-K tensors back to a separate statistics pass, shows here
    print(f"UNet pass at B={B}: {pipe.unet.graph_nodes} launches (hipGraph nodes)")
    assert 0 < pipe.unet.graph_nodes <= 420
    sp_nf = s1.check_finite()   # the bench configuration stays inside fp16's range (non-finite detector of both graphs)
    for o in outs:
        for k in ("latents", "features", "rgb"):
            assert torch.equal(o[k], serial[k]), f"pipelined batch differs from the serial run in {k}"
    W, bias = _probe_head(6, N, 5)
    mask = probe_argmax_mask(outs[-1]["features"], W.to(DEV) * 255.0, bias.to(DEV)).cpu().numpy()
    sel = [0, 7]
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    opipe = op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg))
    ref = op.sample_v6(opipe, x[sel], ctx, N)
    e = rel_err(outs[-1]["latents"][sel], ref["latents"][-1])
    fd = np.abs(outs[-1]["features"][sel].cpu().numpy().astype(int) - ref["features"].astype(int))
    rmask = noise_post.probe_argmax(ref["features"], (W * 255.0).numpy(), bias.numpy())
    ndiff = int((mask[sel] != rmask).sum())
    print(f"configs[1] bench mode B={B}, two batches in flight: final latents of patches {sel}: {err_report(outs[-1]['latents'][sel], ref['latents'][-1])}; "
          f"luma max diff {fd.max()} (!=0: {(fd > 0).mean():.4f}); masks: {ndiff} of {rmask.size} pixels differ")
    assert e <= 1e-3, "north-star tolerance: latents within 1e-3 of the reference (relative to the latent range)"
    assert fd.max() <= 1
    assert assert_mask_flips_within_margin(mask[sel], rmask, ref["features"], W, bias, outs[-1]["features"][sel].cpu().numpy(), "configs[1] bench mode") == ndiff
    assert ndiff <= MASK_FLIP_BOUND, f"{ndiff} mask pixels differ from the oracle's (measured 5-11 in rounds 4-5): regression of the off-by-one rate"


@pytest.mark.timeout(2400)
def test_one_pass_1024_sd15_width_against_oracle():
    """SURVEY 8f row 1 at the reference's real size (segmentor.py:505-530): B = 1, 1024x1024, ONE pass, SD-v1.5 width -- 128x128 latents, the
    level-0 self-attention over 16,384 tokens x 8 heads x d = 40 at width 320 -- against the fp32 CPU oracle."""
    ucfg, vcfg = configs.SD15_UNET, configs.SD15_VAE
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    g = torch.Generator().manual_seed(88)
    x = torch.rand((1, 3, 1024, 1024), generator=g)
    ctx = torch.randn((1, 6, 768), generator=g) * 0.5
    pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, DEV), UNet2DConditionModel(ucfg, usd, DEV))
    out = LaplaceSampler(pipe).sample(x.to(DEV), ctx.to(DEV), 1)
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    ref = op.sample_one_pass(op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg)), x, ctx)
    e = rel_err(out["latents"], ref["latents"])
    rd = np.abs(out["rgb"].cpu().numpy().astype(int) - ref["rgb_u8"].astype(int))
    print(f"1024^2 one pass at SD-v1.5 width: latents {err_report(out['latents'], ref['latents'])}; rgb max diff {rd.max()} (!=0: {(rd > 0).mean():.4f})")
    assert out["rgb"].shape[1:3] == (1024, 1024)
    assert e <= 1e-3 and rd.max() <= 1


def test_pixel_csv_from_device_features(tiny, tmp_path):
    """SURVEY 8f row 3 on the device tensor: sample()["features"][b] (uint8, on the GPU) straight into write_pixel_csv, against the file the
    reference's loop (pixel_latent_vector.py:89-101: dict keyed by (i, j), csv.writer) writes for the same features."""
    import csv
    from ldiffusion_amd import pixel_latent_vector as plv
    g = torch.Generator().manual_seed(31)
    x = torch.rand((2, 3, 64, 64), generator=g)
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    out = LaplaceSampler(tiny["pipe"]).sample(x.to(DEV), ctx.to(DEV), 5)
    label = torch.randint(0, 6, (64, 64), generator=g)
    feats = out["features"][1]
    assert feats.is_cuda and feats.dtype == torch.uint8
    plv.write_pixel_csv(tmp_path / "dev.csv", feats, label.to(DEV))
    f = feats.cpu().numpy()
    pixel_dict = {}
    for i in range(64):
        for j in range(64):
            pixel_dict[(i, j)] = [int(f[k, i, j]) for k in range(5)] + [int(label[i, j])]
    with open(tmp_path / "ref.csv", "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(plv.generate_title(5))
        for key, values in pixel_dict.items():
            wr.writerow([key] + values)
    assert (tmp_path / "dev.csv").read_bytes() == (tmp_path / "ref.csv").read_bytes()


def test_pixel_latent_vector_entry_point_mirror(tiny, tmp_path):
    """Entry point B (pixel_latent_vector.py:58-102) end to end on the device: a loader of batch-size-1 (image, label) pairs -> N-pass sampler in
    device batches -> one CSV per image, against the oracle's V6 loop run image by image (features within one grey level; rows, header and
    label column exact; a batched run and a one-image-at-a-time run -- other split-K plans, i.e. another fp32 summation order -- agree within one
    grey level)."""
    import csv
    from ldiffusion_amd import pixel_latent_vector as plv
    g = torch.Generator().manual_seed(41)
    N = 5
    data = [(torch.rand((1, 3, 64, 64), generator=g), torch.randint(0, 6, (1, 1, 64, 64), generator=g)) for _ in range(5)]
    ctx = torch.randn((1, 6, 64), generator=g) * 0.5
    files = plv.pixel_latent_vector(tiny["pipe"], tiny["vae"], tiny["unet"], N, train_loader=data, text_embeddings=ctx, out_dir=str(tmp_path / "a"), batch_size=3)
    single = plv.pixel_latent_vector(tiny["pipe"], None, None, N, train_loader=data, text_embeddings=ctx, out_dir=str(tmp_path / "b"), batch_size=1)
    assert [os.path.basename(f) for f in files] == [f"pixel_dict_{i}.csv" for i in range(5)]
    for fa, fb in zip(files, single):
        ta, tb = (np.array([[int(v) for v in r[1:]] for r in list(csv.reader(open(f, newline="")))[1:]]) for f in (fa, fb))
        assert ta.shape == tb.shape and np.abs(ta - tb).max() <= 1 and np.array_equal(ta[:, N], tb[:, N])
    for i in (0, 4):
        ref = op.sample_v6(tiny["opipe"], data[i][0], ctx, N)["features"][0]                      # [N, H, W] uint8
        rows = list(csv.reader(open(files[i], newline="")))
        assert rows[0] == plv.generate_title(N) and len(rows) == 1 + 64 * 64
        assert rows[1 + 64 * 3 + 7][0] == "(3, 7)"
        tab = np.array([[int(v) for v in r[1:]] for r in rows[1:]])
        assert np.array_equal(tab[:, N], data[i][1][0, 0].numpy().reshape(-1))
        d = np.abs(tab[:, :N].T.reshape(N, 64, 64) - ref.astype(int))
        assert d.max() <= 1, f"image {i}: features more than one grey level from the oracle"
    with pytest.raises(ValueError):
        plv.pixel_latent_vector(tiny["pipe"], None, None, 2, train_loader=data, text_embeddings=ctx, out_dir=str(tmp_path / "c"))
    with pytest.raises(RuntimeError):
        plv.pixel_latent_vector(tiny["pipe"], None, None, 5, train_loader=None, text_embeddings=ctx)


@pytest.mark.timeout(900)
def test_bench_two_ranks_control_flow_on_one_gpu():
    """bench.py's N > 1 path end to end on a one-GPU box: `python bench.py --gpus 2` starts its two ranks itself (torch.distributed.run, 127.0.0.1), both on
    cuda:0 over gloo (LDIFF_BENCH_SHARED_GPU=1: test only -- the real run is one rank per GPU over RCCL), reduced-width graph.  Checks the contract of the
    line: patch sharding, barrier + max over ranks, the mask all-gather, the serial-vs-pipelined result check, the non-finite check, rank 0 alone printing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LDIFF_BENCH_SHARED_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tiny", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["checked"] is True and d["check"]["finite"] is True
    assert d["comm"]["world_size"] == 2 and d["config"]["parallelism"].startswith("dp2") and d["value"] > 0
