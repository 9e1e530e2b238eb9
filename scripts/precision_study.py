"""CPU study (no GPU): where does the fp16 error of the HIP path come from, and which storage policy meets 1e-3?

Runs the oracle's UNet / VAE graphs with roundings injected at the places the HIP executors round
(`ldiffusion_amd/csrc/model.hip`), under several policies, and prints max|err| / max|ref| against the fp32 oracle.
TEST INFRASTRUCTURE: imports oracle/, never imported by the product.

Policies (cumulative):
  A  everything fp16 in HBM (round 1): weights, every contraction operand, every stored tensor incl. the residual stream
  B  A + fp16-representable checkpoint (no weight-rounding term in the comparison)
  C  B + residual stream kept fp32 (operands still see its fp16 rounding)
  D  C + GroupNorm statistics/apply and LayerNorm read the fp32 stream (single rounding of the normalised operand)
  E  D + branch-internal tensors (conv1 output, q/k/v, attention out, ff) kept fp32 until they become an MFMA operand
"""
import argparse
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from ldiffusion_amd import configs, weights  # noqa: E402
from oracle import unet as ou, vae as ov  # noqa: E402


def h(x):
    return x.to(torch.float16).to(torch.float32)


class Pol:
    def __init__(self, w16=True, stream32=False, norm32=False, inner32=False, split=False, split_all=False):
        self.w16, self.stream32, self.norm32, self.inner32, self.split, self.split_all = w16, stream32, norm32, inner32, split, split_all

    def W(self, w):
        return h(w) if self.w16 else w

    def S(self, x):   # residual stream storage
        return x if self.stream32 else h(x)

    def I(self, x):   # branch-internal storage
        return x if self.inner32 else h(x)

    def N(self, x):   # what a norm reads from the stream
        return x if self.norm32 else h(x)


STREAM_OPS = ("conv_shortcut", "proj_in", "proj_out", "samplers", "conv_in", "quant_conv")


def conv(P, sd, p, x, stride=1, padding=1):
    a = x if (P.split_all or (P.split and any(t in p for t in STREAM_OPS))) else h(x)   # split: operand carried as fp16 hi + fp16 lo (K doubled)
    return F.conv2d(a, P.W(sd[p + ".weight"]), sd[p + ".bias"], stride=stride, padding=padding)


def lin(P, sd, p, x):
    a = x if (P.split_all and p.endswith(("to_q", "to_k", "to_v", "query", "key", "value"))) else h(x)   # split_all: GN-applied operand carried hi|lo
    return F.linear(a, P.W(sd[p + ".weight"]), sd.get(p + ".bias"))


def gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet(P, sd, p, x, temb, groups, eps):
    a = F.silu(gn(sd, p + ".norm1", P.N(x), groups, eps))
    t = conv(P, sd, p + ".conv1", a)
    if temb is not None:
        t = t + F.linear(h(F.silu(temb)), P.W(sd[p + ".time_emb_proj.weight"]), sd[p + ".time_emb_proj.bias"])[:, :, None, None]
    t = P.I(t)
    a = F.silu(gn(sd, p + ".norm2", t, groups, eps))
    t = conv(P, sd, p + ".conv2", a)
    if (p + ".conv_shortcut.weight") in sd:
        x = conv(P, sd, p + ".conv_shortcut", x, padding=0)   # fp32 accumulate, added before the store
    return P.S(x + t)


def attn(P, sd, p, x, ctx, heads):
    ctx = x if ctx is None else ctx
    q, k, v = P.I(lin(P, sd, p + ".to_q", x)), P.I(lin(P, sd, p + ".to_k", ctx)), P.I(lin(P, sd, p + ".to_v", ctx))
    B, Lq, C = q.shape
    d = C // heads
    q = h(q).view(B, Lq, heads, d).transpose(1, 2)
    k = h(k).view(B, -1, heads, d).transpose(1, 2)
    v = h(v).view(B, -1, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / d ** 0.5
    pr = torch.softmax(s, -1)
    # flash kernel: P rounded to fp16 for the PV MFMA, row sum in fp32 (unnormalised exp / max); emulate by rounding the normalised P
    o = h(pr) @ v
    o = P.I(o.transpose(1, 2).reshape(B, Lq, C))
    return lin(P, sd, p + ".to_out.0", o)


def tblock(P, sd, p, x, ctx, heads):
    C = x.shape[-1]
    ln = lambda n, t: F.layer_norm(P.N(t), (C,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)
    x = P.S(attn(P, sd, p + ".attn1", ln("norm1", x), None, heads) + x)
    x = P.S(attn(P, sd, p + ".attn2", ln("norm2", x), ctx, heads) + x)
    f = lin(P, sd, p + ".ff.net.0.proj", ln("norm3", x))
    a, gate = f.chunk(2, dim=-1)
    f = P.I(a * F.gelu(gate))
    return P.S(lin(P, sd, p + ".ff.net.2", f) + x)


def t2d(P, sd, p, x, ctx, heads, groups):
    B, C, H, W = x.shape
    t = gn(sd, p + ".norm", P.N(x), groups, 1e-6)
    t = P.S(conv(P, sd, p + ".proj_in", t, padding=0))
    t = t.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = tblock(P, sd, p + ".transformer_blocks.0", t, ctx, heads)
    t = t.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    return P.S(conv(P, sd, p + ".proj_out", t, padding=0) + x)


def unet(P, sd, cfg, sample, timestep, ctx):
    B = sample.shape[0]
    ctx = ctx.expand(B, -1, -1) if ctx.shape[0] != B else ctx
    boc, groups, eps, heads, lpb = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["norm_eps"], cfg["attention_head_dim"], cfg["layers_per_block"]
    t = torch.as_tensor(timestep)[None].expand(B)
    temb = ou.timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    temb = lin(P, sd, "time_embedding.linear_2", F.silu(lin(P, sd, "time_embedding.linear_1", temb)))
    x = P.S(conv(P, sd, "conv_in", sample))
    skips = [x]
    for i, bt in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            x = resnet(P, sd, f"down_blocks.{i}.resnets.{j}", x, temb, groups, eps)
            if bt == "CrossAttnDownBlock2D":
                x = t2d(P, sd, f"down_blocks.{i}.attentions.{j}", x, ctx, heads, groups)
            skips.append(x)
        if i != len(boc) - 1:
            x = P.S(conv(P, sd, f"down_blocks.{i}.downsamplers.0.conv", x, stride=2))
            skips.append(x)
    x = resnet(P, sd, "mid_block.resnets.0", x, temb, groups, eps)
    x = t2d(P, sd, "mid_block.attentions.0", x, ctx, heads, groups)
    x = resnet(P, sd, "mid_block.resnets.1", x, temb, groups, eps)
    for i, bt in enumerate(cfg["up_block_types"]):
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], 1)
            x = resnet(P, sd, f"up_blocks.{i}.resnets.{j}", x, temb, groups, eps)
            if bt == "CrossAttnUpBlock2D":
                x = t2d(P, sd, f"up_blocks.{i}.attentions.{j}", x, ctx, heads, groups)
        if i != len(boc) - 1:
            x = P.S(conv(P, sd, f"up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest")))
    x = F.silu(gn(sd, "conv_norm_out", P.N(x), groups, eps))
    return conv(P, sd, "conv_out", x)


def vae_attn(P, sd, p, x, groups):
    B, C, H, W = x.shape
    t = F.group_norm(P.N(x).view(B, C, H * W), groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], 1e-6).transpose(1, 2)
    q, k, v = (h(P.I(lin(P, sd, p + "." + n, t))) for n in ("to_q", "to_k", "to_v"))
    pr = torch.softmax((q @ k.transpose(-1, -2)) / C ** 0.5, -1)
    o = P.I(h(pr) @ v)
    o = lin(P, sd, p + ".to_out.0", o)
    return P.S(o.transpose(1, 2).reshape(B, C, H, W) + x)


def vae_mid(P, sd, p, x, groups):
    x = resnet(P, sd, p + ".resnets.0", x, None, groups, 1e-6)
    x = vae_attn(P, sd, p + ".attentions.0", x, groups)
    return resnet(P, sd, p + ".resnets.1", x, None, groups, 1e-6)


def vae_encode(P, sd, cfg, x):
    groups, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    x = P.S(conv(P, sd, "encoder.conv_in", x))
    for i in range(len(boc)):
        for j in range(lpb):
            x = resnet(P, sd, f"encoder.down_blocks.{i}.resnets.{j}", x, None, groups, 1e-6)
        if i != len(boc) - 1:
            x = P.S(conv(P, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(x, (0, 1, 0, 1)), stride=2, padding=0))
    x = vae_mid(P, sd, "encoder.mid_block", x, groups)
    x = F.silu(gn(sd, "encoder.conv_norm_out", P.N(x), groups, 1e-6))
    x = P.I(conv(P, sd, "encoder.conv_out", x))
    return conv(P, sd, "quant_conv", x, padding=0)


def vae_decode(P, sd, cfg, z):
    groups, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    x = P.I(conv(P, sd, "post_quant_conv", z, padding=0))
    x = P.S(conv(P, sd, "decoder.conv_in", x))
    x = vae_mid(P, sd, "decoder.mid_block", x, groups)
    for i in range(len(boc)):
        for j in range(lpb + 1):
            x = resnet(P, sd, f"decoder.up_blocks.{i}.resnets.{j}", x, None, groups, 1e-6)
        if i != len(boc) - 1:
            x = P.S(conv(P, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest")))
    x = F.silu(gn(sd, "decoder.conv_norm_out", P.N(x), groups, 1e-6))
    return conv(P, sd, "decoder.conv_out", x)


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).pow(2).mean().sqrt() / b.abs().max()).item()


POLICIES = {
    "A all-fp16 (round 1), fp32 checkpoint": (dict(), False),
    "B all-fp16, fp16 checkpoint": (dict(), True),
    "C + fp32 residual stream": (dict(stream32=True), True),
    "D + norms read the fp32 stream": (dict(stream32=True, norm32=True), True),
    "E + fp32 branch internals": (dict(stream32=True, norm32=True, inner32=True), True),
    "F C + split (hi|lo) stream operands": (dict(stream32=True, split=True), True),
    "G F + norms read hi+lo": (dict(stream32=True, split=True, norm32=True), True),
    "S every conv operand split (encoder policy)": (dict(stream32=True, split=True, norm32=True, inner32=True, split_all=True), True),
    "H G + fp32 branch internals": (dict(stream32=True, split=True, norm32=True, inner32=True), True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", choices=["tiny", "sd15"], default="tiny")
    ap.add_argument("--hw", type=int, default=32)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--vae", action="store_true")
    a = ap.parse_args()
    ucfg, vcfg = (configs.TINY_UNET, configs.TINY_VAE) if a.width == "tiny" else (configs.SD15_UNET, configs.SD15_VAE)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(2)
    x = torch.randn((a.B, 4, a.hw, a.hw), generator=g)
    ctx = torch.randn((1, 6, ucfg["cross_attention_dim"]), generator=g) * 0.5
    img = torch.rand((a.B, 3, a.hw * 8, a.hw * 8), generator=g)
    usd32 = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42)
    vsd32 = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43)
    usd16 = {k: h(v) for k, v in usd32.items()}
    vsd16 = {k: h(v) for k, v in vsd32.items()}
    with torch.no_grad():
        for name, (kw, ck16) in POLICIES.items():
            P = Pol(**kw)
            usd, vsd = (usd16, vsd16) if ck16 else (usd32, vsd32)
            t0 = time.time()
            ref = ou.unet_forward(usd, ucfg, x, 501, ctx).sample
            got = unet(P, usd, ucfg, x, 501, ctx)
            line = f"{name:42s} unet max {rel(got, ref)[0]:.2e} rms {rel(got, ref)[1]:.2e}"
            if a.vae:
                rm = ov.vae_encode_moments(vsd, vcfg, img)[:, :4]
                gm = vae_encode(P, vsd, vcfg, img)[:, :4]
                rd = ov.vae_decode(vsd, vcfg, x * 0.5)
                gd = vae_decode(P, vsd, vcfg, x * 0.5)
                line += f" | enc max {rel(gm, rm)[0]:.2e} rms {rel(gm, rm)[1]:.2e} | dec max {rel(gd, rd)[0]:.2e} rms {rel(gd, rd)[1]:.2e}"
            print(line + f"  ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
