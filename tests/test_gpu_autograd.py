"""-m gpu: the backward-pass primitives (include/ldiff.h "Backward-pass primitives", ldiffusion_amd/autograd.py) against torch.autograd
on the CPU, in float32, on the same fp16-rounded inputs.

Tolerance: activations, their gradients and the MFMA operands are float16 with float32 accumulation; parameter gradients are float32
sums of float16 products.  |got - ref| <= 4e-3 * max|ref| for every output and gradient (measured values are printed)."""
import math

import pytest
import torch
import torch.nn.functional as F

from ldiffusion_amd import autograd as ag

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 4e-3


def r16(x):
    return x.to(torch.float16).to(torch.float32)


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-2)).item()   # (a gradient that is exactly 0 in the reference: absolute 4e-5)


CONV_CASES = {
    # name: (B, Cin, H, W, Cout, k, stride, ups, bias)
    "3x3": (2, 64, 8, 8, 128, 3, 1, 0, True),
    "3x3_stride2": (2, 64, 8, 8, 64, 3, 2, 0, True),
    "3x3_upsample": (2, 64, 4, 4, 64, 3, 1, 1, True),
    "1x1_shortcut": (2, 192, 8, 8, 64, 1, 1, 0, True),
    "3x3_cout4_conv_out": (2, 64, 8, 8, 4, 3, 1, 0, True),
    "3x3_cin4_conv_in": (2, 4, 8, 8, 64, 3, 1, 0, True),
    "3x3_ragged_M": (1, 64, 3, 5, 72, 3, 1, 0, False),
    "3x3_wide_map": (1, 32, 32, 32, 32, 3, 1, 0, True),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv_forward_dgrad_wgrad(name):
    B, Cin, H, W, Cout, k, stride, ups, has_bias = CONV_CASES[name]
    g = torch.Generator().manual_seed(len(name) * 7 + Cout)
    x = r16(torch.randn((B, Cin, H, W), generator=g))
    w = r16(torch.randn((Cout, Cin, k, k), generator=g) / math.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g) * 0.1 if has_bias else None
    # reference
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    xin = F.interpolate(xr, scale_factor=2.0, mode="nearest") if ups else xr
    yr = F.conv2d(xin, wr, br, stride=stride, padding=k // 2)
    G = r16(torch.randn(yr.shape, generator=g))
    (yr * G).sum().backward()
    # device: NHWC f16, channels padded to 8
    Cp = (Cin + 7) // 8 * 8
    xd = torch.zeros((B, H, W, Cp), dtype=torch.float16)
    xd[..., :Cin] = x.permute(0, 2, 3, 1).to(torch.float16)
    xd = xd.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if has_bias else None
    y = ag.Conv2dFn.apply(xd, wd, bd, stride, ups)
    Gd = torch.zeros(y.shape, dtype=torch.float16)
    Gd[..., :Cout] = G.permute(0, 2, 3, 1).to(torch.float16)
    (y.float() * Gd.to(DEV).float()).sum().backward()
    e_y = rel(y[..., :Cout].permute(0, 3, 1, 2), yr)
    e_dx = rel(xd.grad[..., :Cin].permute(0, 3, 1, 2), xr.grad)
    e_dw = rel(wd.grad, wr.grad)
    e_db = rel(bd.grad, br.grad) if has_bias else 0.0
    print(f"conv {name}: y {e_y:.2e} dx {e_dx:.2e} dw {e_dw:.2e} db {e_db:.2e}")
    assert max(e_y, e_dx, e_dw, e_db) <= TOL
    assert (y[..., Cout:] == 0).all() and wd.grad.dtype == torch.float32


def test_linear_rows():
    g = torch.Generator().manual_seed(3)
    x = r16(torch.randn((2, 64, 320), generator=g))
    w = r16(torch.randn((960, 320), generator=g) / math.sqrt(320))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.linear(xr, wr)
    G = r16(torch.randn(yr.shape, generator=g))
    (yr * G).sum().backward()
    xd, wd = x.to(torch.float16).to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ag.linear(xd, wd)
    (y.float() * G.to(DEV)).sum().backward()
    e = (rel(y, yr), rel(xd.grad, xr.grad), rel(wd.grad, wr.grad))
    print(f"linear: y {e[0]:.2e} dx {e[1]:.2e} dw {e[2]:.2e}")
    assert max(e) <= TOL


@pytest.mark.parametrize("B,C,H,W,groups,silu", [(2, 64, 8, 8, 32, 1), (2, 320, 8, 8, 32, 1), (1, 128, 16, 16, 32, 0), (3, 96, 2, 2, 32, 1), (2, 1920, 4, 4, 32, 1)])
def test_group_norm_silu_backward(B, C, H, W, groups, silu):
    g = torch.Generator().manual_seed(C + H)
    x = r16(torch.randn((B, C, H, W), generator=g) * 1.5 + 0.3)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.group_norm(xr, groups, gr, br, 1e-5)
    yr = F.silu(yr) if silu else yr
    G = r16(torch.randn(yr.shape, generator=g))
    (yr * G).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    y = ag.GroupNormFn.apply(xd, gd, bd, groups, 1e-5, silu)
    (y.float() * G.permute(0, 2, 3, 1).to(DEV)).sum().backward()
    e = (rel(y.permute(0, 3, 1, 2), yr), rel(xd.grad.permute(0, 3, 1, 2), xr.grad), rel(gd.grad, gr.grad), rel(bd.grad, br.grad))
    print(f"group norm C={C} {H}x{W} silu={silu}: y {e[0]:.2e} dx {e[1]:.2e} dgamma {e[2]:.2e} dbeta {e[3]:.2e}")
    assert max(e) <= TOL


@pytest.mark.parametrize("rows,C", [(128, 320), (7, 1280), (512, 64)])
def test_layer_norm_backward(rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = r16(torch.randn((rows, C), generator=g) * 2 + 0.5)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gr, br, 1e-5)
    G = r16(torch.randn(yr.shape, generator=g))
    (yr * G).sum().backward()
    xd = x.to(torch.float16).to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    y = ag.LayerNormFn.apply(xd, gd, bd, 1e-5)
    (y.float() * G.to(DEV)).sum().backward()
    e = (rel(y, yr), rel(xd.grad, xr.grad), rel(gd.grad, gr.grad), rel(bd.grad, br.grad))
    print(f"layer norm {rows}x{C}: y {e[0]:.2e} dx {e[1]:.2e} dgamma {e[2]:.2e} dbeta {e[3]:.2e}")
    assert max(e) <= TOL


def test_geglu_backward():
    g = torch.Generator().manual_seed(5)
    x = r16(torch.randn((300, 2 * 256), generator=g) * 1.5)
    xr = x.clone().requires_grad_(True)
    h, gate = xr.chunk(2, -1)
    yr = h * F.gelu(gate)
    G = r16(torch.randn(yr.shape, generator=g))
    (yr * G).sum().backward()
    xd = x.to(torch.float16).to(DEV).requires_grad_(True)
    y = ag.GegluFn.apply(xd)
    (y.float() * G.to(DEV)).sum().backward()
    e = (rel(y, yr), rel(xd.grad, xr.grad))
    print(f"geglu: y {e[0]:.2e} dx {e[1]:.2e}")
    assert max(e) <= TOL


def test_silu_backward():
    """ldiff_op_silu / ldiff_op_silu_bwd (the time-embedding MLP's activations of the fine-tuning step) against autograd over F.silu."""
    g = torch.Generator().manual_seed(6)
    x = r16(torch.randn((3, 1280), generator=g) * 3.0)
    xr = x.clone().requires_grad_(True)
    yr = F.silu(xr)
    G = r16(torch.randn(yr.shape, generator=g))
    (yr * G).sum().backward()
    xd = x.to(torch.float16).to(DEV).requires_grad_(True)
    y = ag.silu(xd)
    (y.float() * G.to(DEV)).sum().backward()
    e = (rel(y, yr), rel(xd.grad, xr.grad))
    print(f"silu: y {e[0]:.2e} dx {e[1]:.2e}")
    assert max(e) <= TOL and y.dtype == torch.float16 and xd.grad.dtype == torch.float16


@pytest.mark.parametrize("B,heads,Lq,Lk,d", [(2, 8, 64, 64, 40), (2, 8, 16, 16, 80), (2, 8, 4, 4, 160), (2, 8, 64, 6, 40), (2, 1, 64, 64, 512), (1, 8, 1, 1, 32)])
def test_attention_backward(B, heads, Lq, Lk, d):
    g = torch.Generator().manual_seed(Lq + d)
    Cc = heads * d
    q, k, v = (r16(torch.randn((B, L, Cc), generator=g)) for L in (Lq, Lk, Lk))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    sh = lambda t, L: t.view(B, L, heads, d).transpose(1, 2)
    orf = F.scaled_dot_product_attention(sh(qr, Lq), sh(kr, Lk), sh(vr, Lk)).transpose(1, 2).reshape(B, Lq, Cc)
    G = r16(torch.randn(orf.shape, generator=g))
    (orf * G).sum().backward()
    qd, kd, vd = (t.to(torch.float16).to(DEV).requires_grad_(True) for t in (q, k, v))
    o = ag.AttentionFn.apply(qd, kd, vd, heads)
    (o.float() * G.to(DEV)).sum().backward()
    e = (rel(o, orf), rel(qd.grad, qr.grad), rel(kd.grad, kr.grad), rel(vd.grad, vr.grad))
    print(f"attention B={B} h={heads} {Lq}x{Lk} d={d}: o {e[0]:.2e} dq {e[1]:.2e} dk {e[2]:.2e} dv {e[3]:.2e}")
    assert max(e) <= TOL
    with pytest.raises(ValueError):
        big = torch.zeros((1, 256, 64), dtype=torch.float16, device=DEV, requires_grad=True)
        ag.AttentionFn.apply(big, big, big, 8).sum().backward()       # 256 x 256 scores exceed the short-sequence backward kernel


def test_adamw_matches_torch():
    g = torch.Generator().manual_seed(8)
    ps = [torch.randn(s, generator=g) for s in ((1000,), (64, 33), (5,))]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    dev = [p.clone().to(DEV) for p in ps]
    state = {}
    for _ in range(4):
        grads = [torch.randn(p.shape, generator=g) for p in ps]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
        ag.adamw_step(dev, [gr.to(DEV) for gr in grads], state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for d, r in zip(dev, ref):
        assert (d.cpu() - r.detach()).abs().max() <= 2e-6 * max(1.0, r.abs().max().item())


def test_pack_weight_multi_equals_the_single_launches():
    """ag.PackPlan (ldiff_op_pack_weight_multi: every forward and dgrad layout of a weight list in one launch) against ldiff_op_pack_weight
    tensor by tensor: 3x3 and 1x1 convs, linears, ragged channel counts (padding rows / columns must come out zero)."""
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 4, 3, 3), (4, 64, 3, 3), (320, 320, 3, 3), (130, 70, 3, 3), (640, 320, 1, 1), (96, 200), (1280, 320), (24, 40, 1, 1)]
    ws = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    plan = ag.PackPlan(ws)
    plan.flat.fill_(float("nan"))
    plan.run()
    torch.cuda.synchronize()
    for w in ws:
        for dgrad in (False, True):
            one = ag.pack_weight(w, dgrad=dgrad)
            with ag.packed_weights(plan):
                many = ag.pack_weight(w, dgrad=dgrad)
            assert many.data_ptr() != one.data_ptr() and many.shape == one.shape
            assert torch.equal(many, one), (tuple(w.shape), dgrad)
