"""CPU study (no GPU): which VAE-encoder layers need split (fp16 hi|lo) conv operands for the 1e-3 latent contract?
The product runs the whole encoder at precision 2 (every conv operand split: 2x the MFMA work of its convs, 22.3 against 13.5 ms);
this emulates precision 2 on a subset of the blocks and precision 1 on the rest (scripts/precision_study.py's rounding model).
TEST INFRASTRUCTURE: imports oracle/, never imported by the product.   usage: python scripts/precision_study_encoder.py [--hw 32]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import precision_study as ps  # noqa: E402
from ldiffusion_amd import configs, weights  # noqa: E402
from oracle import vae as ov  # noqa: E402


class PolE(ps.Pol):
    """precision 1 everywhere; precision 2 (split operands, split branch internals) where `full(p)` says so"""

    def __init__(self, full):
        super().__init__(stream32=True, split=True, norm32=True, inner32=False, split_all=False)
        self.full = full


_conv0, _resnet0 = ps.conv, ps.resnet


def conv(P, sd, p, x, stride=1, padding=1):
    if isinstance(P, PolE) and P.full(p):
        return torch.nn.functional.conv2d(x, P.W(sd[p + ".weight"]), sd[p + ".bias"], stride=stride, padding=padding)
    return _conv0(P, sd, p, x, stride, padding)


def resnet(P, sd, p, x, temb, groups, eps):
    if isinstance(P, PolE):
        Q = PolE(P.full)
        Q.inner32 = P.full(p)
        return _resnet0(Q, sd, p, x, temb, groups, eps)
    return _resnet0(P, sd, p, x, temb, groups, eps)


ps.conv, ps.resnet = conv, resnet


def block_of(p):
    """0..3 = encoder.down_blocks.i (and its downsampler), 4 = mid block, 5 = conv_norm_out / conv_out, -1 = conv_in / quant_conv"""
    if "down_blocks." in p:
        return int(p.split("down_blocks.")[1][0])
    if "mid_block" in p:
        return 4
    if "conv_out" in p:
        return 5
    return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=32, help="latent size (image = 8x)")
    a = ap.parse_args()
    vcfg = configs.SD15_VAE
    g = torch.Generator().manual_seed(2)
    img = torch.rand((1, 3, a.hw * 8, a.hw * 8), generator=g)
    vsd = {k: ps.h(v) for k, v in weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43).items()}
    flops = {0: 0.55, 1: 0.29, 2: 0.10, 3: 0.03, 4: 0.03, 5: 0.0}   # share of the encoder's conv MACs per block (B x 512^2: 128ch@512^2 dominates)
    with torch.no_grad():
        ref = ov.vae_encode_moments(vsd, vcfg, img)[:, :4]
        cases = [("precision 1 everywhere", set()), ("precision 2 everywhere", {0, 1, 2, 3, 4, 5})]
        cases += [(f"precision 2 on blocks >= {k} (down_blocks.{k}.., mid, conv_out)", set(range(k, 6))) for k in (1, 2, 3, 4)]
        cases += [(f"precision 2 on blocks <= {k}", set(range(0, k + 1))) for k in (0, 1)]
        cases += [(f"precision 2 on block {k} only", {k}) for k in range(6)]
        for name, blocks in cases:
            t0 = time.time()
            got = ps.vae_encode(PolE(lambda p, b=blocks: block_of(p) in b), vsd, vcfg, img)[:, :4]
            mx, rms = ps.rel(got, ref)
            extra = sum(flops[b] for b in blocks)
            print(f"{name:64s} enc mean: max {mx:.2e} rms {rms:.2e} of range;  split MACs {extra:.2f} of the encoder's ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
