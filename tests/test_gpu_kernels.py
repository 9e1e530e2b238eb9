"""-m gpu: every HIP kernel through the C ABI against plain torch CPU ops on the same fp16-rounded inputs.

Tolerance (floating-point path, stated per the task contract): operands are fp16, accumulation fp32, outputs
rounded to fp16 (rel. 2^-11), so |got - ref| <= 2e-3*max|ref| + 2e-3*|ref| unless a test says otherwise.
Integer outputs (uint8 images, luma, argmax) are compared bit-exactly given identical float inputs.
"""
import ctypes as C
import math
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ldiffusion_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def nhwc16(x):  # [B,C,H,W] f32 cpu -> [B,H,W,C] f16 cuda
    return x.permute(0, 2, 3, 1).contiguous().to(torch.float16).to(DEV)


def r16(x):  # fp16-rounded copy kept in fp32 (what the kernel actually sees)
    return x.to(torch.float16).to(torch.float32)


def assert_close(got, ref, what, rtol=2e-3, atol_rel=2e-3):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs()
    tol = atol_rel * ref.abs().max() + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {err.max():.4e} (max|ref| {ref.abs().max():.3f})"


def run_conv(lib, x, w, bias=None, x2=None, stride=1, pad=(1, 1), ups=0, gn=None, silu=0, temb=None, res=None, out_f32=False,
             asym=False):
    """x,x2: [B,C,H,W] f32 cpu; w: [Cout,Cin,ks,ks].  Returns ([B,Cout,Ho,Wo] got, ref)."""
    B, C1, H, W = x.shape
    C2 = x2.shape[1] if x2 is not None else 0
    Cout, Cin, ks, _ = w.shape
    assert Cin == C1 + C2
    xin = r16(x) if x2 is None else torch.cat([r16(x), r16(x2)], 1)
    # ---- reference ----
    a = xin
    if gn is not None:
        scale, shift = gn  # [B, Cin]
        a = a * scale[:, :, None, None] + shift[:, :, None, None]
        if silu:
            a = F.silu(a)
        a = r16(a)  # the kernel rounds the normalised operand to fp16 before the MFMA
    if ups:
        a = F.interpolate(a, scale_factor=2.0, mode="nearest")
    wr = r16(w)
    if asym:
        a = F.pad(a, (0, 1, 0, 1))
        ref = F.conv2d(a, wr, None, stride=stride, padding=0)
    else:
        ref = F.conv2d(a, wr, None, stride=stride, padding=pad[0])
    if bias is not None:
        ref = ref + bias[None, :, None, None]
    if temb is not None:
        ref = ref + temb[:, :, None, None]
    if res is not None:
        ref = ref + r16(res)
    Ho, Wo = ref.shape[2], ref.shape[3]
    # ---- device ----
    Nrows = (Cout + 15) // 16 * 16
    Nst = (Cout + 3) // 4 * 4
    wd = torch.zeros((Nrows, ks, ks, Cin), dtype=torch.float16)
    wd[:Cout] = w.permute(0, 2, 3, 1).to(torch.float16)
    wd = wd.reshape(Nrows, ks * ks * Cin).contiguous().to(DEV)
    keep = [wd]
    a_ = _lib.ConvArgs()
    xd = nhwc16(x); keep.append(xd)
    a_.x = xd.data_ptr(); a_.C1 = C1
    if x2 is not None:
        x2d = nhwc16(x2); keep.append(x2d)
        a_.x2 = x2d.data_ptr(); a_.C2 = C2
    a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = B, H, W, Ho, Wo
    a_.ks, a_.stride, a_.ups = ks, stride, ups
    a_.pad_t, a_.pad_l = (0, 0) if asym else pad
    a_.w = wd.data_ptr(); a_.N = Nst; a_.Nrows = Nrows; a_.n_real = Cout
    if gn is not None:
        sd, hd = gn[0].contiguous().to(DEV), gn[1].contiguous().to(DEV); keep += [sd, hd]
        a_.gn_scale, a_.gn_shift, a_.silu_in = sd.data_ptr(), hd.data_ptr(), silu
    if bias is not None:
        bd = torch.zeros(Nrows); bd[:Cout] = bias; bd = bd.to(DEV); keep.append(bd)
        a_.bias = bd.data_ptr()
    if temb is not None:
        ld = (Cout + 3) // 4 * 4
        td = torch.zeros((B, ld)); td[:, :Cout] = temb; td = td.to(DEV); keep.append(td)
        a_.temb, a_.ld_temb = td.data_ptr(), ld
    if res is not None:
        rd = torch.zeros((B, Ho, Wo, Nst), dtype=torch.float16); rd[..., :Cout] = res.permute(0, 2, 3, 1).to(torch.float16)
        rd = rd.to(DEV); keep.append(rd)
        a_.res, a_.ld_res = rd.data_ptr(), Nst
    y = torch.full((B, Ho, Wo, Nst), float("nan"), dtype=torch.float32 if out_f32 else torch.float16, device=DEV)
    a_.y, a_.ldy, a_.out_f32 = y.data_ptr(), Nst, int(out_f32)
    _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    torch.cuda.synchronize()
    got = y[..., :Cout].permute(0, 3, 1, 2).float().cpu()
    return got, ref


CONV_CASES = {
    # name: (B, C1, C2, H, W, Cout, ks, stride, ups, asym, gn, extras)
    "3x3_64_64": (2, 64, 0, 16, 16, 64, 3, 1, 0, False, False, False),
    "3x3_320_320_tile128x64": (2, 320, 0, 128, 96, 320, 3, 1, 0, False, False, True),
    "3x3_128_256_tile128x128": (8, 128, 0, 64, 64, 256, 3, 1, 0, False, True, True),
    "1x1_linear_tailM": (1, 320, 0, 1, 300, 960, 1, 1, 0, False, False, False),
    "3x3_stride2_sym": (2, 64, 0, 16, 16, 128, 3, 2, 0, False, False, False),
    "3x3_stride2_asym_vae": (2, 64, 0, 16, 16, 64, 3, 2, 0, True, False, False),
    "3x3_upsample2x": (2, 128, 0, 8, 8, 128, 3, 1, 1, False, False, True),
    "3x3_concat_128_64_gn": (2, 128, 64, 16, 16, 128, 3, 1, 0, False, True, True),
    "1x1_concat_shortcut": (2, 128, 64, 16, 16, 64, 1, 1, 0, False, False, False),
    "3x3_cin8_general_path": (2, 8, 0, 32, 32, 320, 3, 1, 0, False, False, False),
    "3x3_cin32_general_path_gn": (2, 32, 0, 16, 16, 32, 3, 1, 0, False, True, True),
    "3x3_cout4_f32": (2, 320, 0, 16, 16, 4, 3, 1, 0, False, True, False),
    "3x3_cout3_f32": (1, 128, 0, 32, 32, 3, 3, 1, 0, False, True, False),
    # narrow-output kernel (N == 4 stored columns, weights resident in LDS): ragged borders, several slabs, two sources, no GroupNorm
    "3x3_cout3_f32_ragged_20x27": (2, 128, 0, 20, 27, 3, 3, 1, 0, False, True, False),
    "3x3_cout4_f32_concat_320": (2, 192, 128, 24, 40, 4, 3, 1, 0, False, True, False),
    "3x3_cout3_f32_plain_512": (1, 512, 0, 16, 48, 3, 3, 1, 0, False, False, False),
    "3x3_cout4_f32_large": (8, 128, 0, 128, 128, 4, 3, 1, 0, False, True, False),
    # <= 3 output channels over 128 input channels: the tap-folded kernel (conv3x3nt_kernel); 2800 tiles on 512 workgroups: 5 or 6 each, ragged borders
    "3x3_cout3_f32_several_tiles_per_workgroup_ragged": (8, 128, 0, 200, 216, 3, 3, 1, 0, False, True, False),
    "3x3_cout3_f32_plain_128_ragged": (2, 128, 0, 20, 27, 3, 3, 1, 0, False, False, False),
    "3x3_cout2_f32_gn_128": (1, 128, 0, 8, 16, 2, 3, 1, 0, False, True, False),
    "1x1_cin8_cout8": (2, 8, 0, 8, 8, 8, 1, 1, 0, False, False, False),
    "3x3_1x1_spatial": (3, 64, 0, 1, 1, 64, 3, 1, 0, False, False, False),
    "3x3_splitk_8x8_1280": (2, 1280, 0, 8, 8, 256, 3, 1, 0, False, True, True),
    "3x3_splitk_concat_16x16": (1, 640, 640, 16, 16, 128, 3, 1, 0, False, True, True),
    # stride-2 convs with few output tiles and a long K loop: split-K in the register-staged implicit GEMM (igemm_splitk_plan)
    "3x3_stride2_splitk_512_16x16": (2, 512, 0, 16, 16, 256, 3, 2, 0, False, False, True),
    "3x3_stride2_splitk_concat_asym": (1, 256, 256, 16, 16, 192, 3, 2, 0, True, False, True),
    "3x3_stride2_splitk_gn": (2, 640, 0, 8, 8, 128, 3, 2, 0, False, True, True),
    # ... and in the LDS-DMA GEMM (concat shortcut conv of the 8x8 level: M = 512, K = 2560)
    "1x1_splitk_concat_2560": (8, 1280, 1280, 8, 8, 256, 1, 1, 0, False, False, True),
    "1x1_splitk_4096_tailM": (1, 4096, 0, 1, 300, 192, 1, 1, 0, False, False, True),
    "1x1_splitk_128x128_tiles": (1, 2560, 0, 1, 1000, 640, 1, 1, 0, False, False, True),   # 8 x 5 tiles of 128 x 128, 40 K-steps -> S = 2
    # two concat sources of UNEQUAL width, so that a K split starts strictly inside the second source (its row pitch differs from the first's)
    "1x1_splitk_concat_1024_2048": (8, 1024, 2048, 8, 8, 256, 1, 1, 0, False, False, True),
    "1x1_splitk_concat_1280_960": (8, 1280, 960, 8, 8, 256, 1, 1, 0, False, False, True),
    "1x1_splitk_concat_1536_768": (8, 1536, 768, 8, 8, 192, 1, 1, 0, False, False, False),
    # wide-tile kernel (8x16 pixel tiles) on maps that are not multiples of the tile, with and without the 16-byte store path
    "3x3_wide_ragged_20x27_gn": (2, 64, 0, 20, 27, 96, 3, 1, 0, False, True, True),
    "3x3_wide_ragged_upsample_9x11": (2, 128, 0, 9, 11, 64, 3, 1, 1, False, False, True),
    "3x3_wide_cout20_8byte_stores": (1, 128, 0, 24, 40, 20, 3, 1, 0, False, True, True),
    "3x3_wide_160_tile_ragged": (1, 64, 0, 17, 33, 320, 3, 1, 0, False, True, False),
    "1x1_linear_N328_16byte_tail": (1, 64, 0, 1, 200, 328, 1, 1, 0, False, False, True),
    # large maps (>= 512 workgroups of the 8x16-tile kernel), several 64-channel slabs, ragged widths, parity folding over 3 slabs
    "3x3_x16_128_128_gn": (8, 128, 0, 128, 128, 128, 3, 1, 0, False, True, True),
    "3x3_x16_256_128_two_slab_pairs": (8, 256, 0, 128, 128, 128, 3, 1, 0, False, False, True),
    "3x3_x16_ragged_width_gn": (8, 128, 0, 128, 120, 128, 3, 1, 0, False, True, True),
    "3x3_x16_concat_bn64_gn": (8, 64, 64, 128, 128, 64, 3, 1, 0, False, True, True),
    "3x3_x16_upsample_parity": (8, 64, 0, 64, 64, 64, 3, 1, 1, False, False, True),
    "3x3_x16_upsample_parity_192ch": (8, 192, 0, 64, 64, 128, 3, 1, 1, False, False, False),
    "3x3_x16_single_slab": (8, 64, 0, 128, 128, 128, 3, 1, 0, False, True, False),
    # persistent 16x16-tile kernel (kernels_conv3x3p.hip: no GroupNorm prologue, tile lists that fill the chip evenly): two tiles per
    # workgroup, exactly one, ragged tiles on every border, 64- and 128-channel n-tiles, two n-tiles, parity folding
    "3x3_p16_128_128_res": (8, 128, 0, 128, 128, 128, 3, 1, 0, False, False, True),
    "3x3_p16_one_tile_per_workgroup": (1, 128, 0, 256, 256, 128, 3, 1, 0, False, False, True),
    "3x3_p16_ragged_72x88": (8, 64, 0, 72, 88, 128, 3, 1, 0, False, False, True),
    "3x3_p16_bn64_concat": (8, 64, 64, 128, 128, 64, 3, 1, 0, False, False, False),
    "3x3_p16_two_ntiles_256": (4, 64, 0, 128, 128, 256, 3, 1, 0, False, False, True),
    "3x3_p16_upsample_parity_ragged": (8, 64, 0, 56, 72, 128, 3, 1, 1, False, False, False),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv(lib, name):
    B, C1, C2, H, W, Cout, ks, stride, ups, asym, use_gn, extras = CONV_CASES[name]
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    Cin = C1 + C2
    x = torch.randn((B, C1, H, W), generator=g)
    x2 = torch.randn((B, C2, H, W), generator=g) if C2 else None
    w = torch.randn((Cout, Cin, ks, ks), generator=g) / math.sqrt(Cin * ks * ks)
    bias = torch.randn(Cout, generator=g) * 0.1
    gn = (1.0 + 0.2 * torch.randn((B, Cin), generator=g), 0.2 * torch.randn((B, Cin), generator=g)) if use_gn else None
    out_f32 = Cout <= 4
    He, We = H << ups, W << ups
    Ho = He // 2 if asym else (He + 2 * (ks // 2) - ks) // stride + 1
    Wo = We // 2 if asym else (We + 2 * (ks // 2) - ks) // stride + 1
    temb = torch.randn((B, Cout), generator=g) * 0.3 if extras else None
    res = torch.randn((B, Cout, Ho, Wo), generator=g) if extras else None
    got, ref = run_conv(lib, x, w, bias, x2, stride, (ks // 2, ks // 2), ups, gn, 1 if use_gn else 0, temb, res, out_f32, asym)
    assert_close(got, ref, name)


@pytest.mark.parametrize("silu", [0, 1])
def test_narrow_conv_tap_folded_against_the_lds_image_kernel(lib, silu):
    """conv_out of the VAE decoder (128 -> 3, GroupNorm prologue): the tap-folded kernel (taps as output columns of one narrow GEMM over the halo
    pixels, then a gather) against torch AND against the LDS-image kernel it replaces (LDIFF_CONV3X3_NARROW_FOLD=0 in a child process is not
    needed: n_real = 0 keeps the launch on the old kernel) -- same operands, a different fp32 summation order: equal to 1e-5 of the output range."""
    g = torch.Generator().manual_seed(11 + silu)
    B, Cc, H, W, Cout = 3, 128, 40, 56, 3
    x = torch.randn((B, Cc, H, W), generator=g)
    w = torch.randn((Cout, Cc, 3, 3), generator=g) / math.sqrt(9 * Cc)
    bias = torch.randn(Cout, generator=g) * 0.1
    gn = (1.0 + 0.2 * torch.randn((B, Cc), generator=g), 0.2 * torch.randn((B, Cc), generator=g))
    got, ref = run_conv(lib, x, w, bias, gn=gn, silu=silu, out_f32=True)
    assert_close(got, ref, f"tap-folded silu={silu}", rtol=1e-4, atol_rel=1e-4)
    # the same launch with n_real = 0 (not stated): the LDS-image kernel
    Nrows = 16
    wd = torch.zeros((Nrows, 3, 3, Cc), dtype=torch.float16); wd[:Cout] = w.permute(0, 2, 3, 1).to(torch.float16)
    wd = wd.reshape(Nrows, 9 * Cc).contiguous().to(DEV)
    xd = nhwc16(x)
    sd, hd = gn[0].contiguous().to(DEV), gn[1].contiguous().to(DEV)
    bd = torch.zeros(Nrows); bd[:Cout] = bias; bd = bd.to(DEV)
    ys = []
    for n_real in (Cout, 0):
        a_ = _lib.ConvArgs()
        a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout, a_.ks, a_.stride, a_.pad_t, a_.pad_l = xd.data_ptr(), Cc, B, H, W, H, W, 3, 1, 1, 1
        a_.w, a_.N, a_.Nrows, a_.n_real, a_.bias = wd.data_ptr(), 4, Nrows, n_real, bd.data_ptr()
        a_.gn_scale, a_.gn_shift, a_.silu_in = sd.data_ptr(), hd.data_ptr(), silu
        y = torch.full((B, H, W, 4), float("nan"), device=DEV)
        a_.y, a_.ldy, a_.out_f32 = y.data_ptr(), 4, 1
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        torch.cuda.synchronize()
        ys.append(y.cpu())
    assert torch.isfinite(ys[0]).all() and torch.equal(ys[0][..., 3], ys[1][..., 3])          # the pad column: the bias of the zero row
    d = (ys[0] - ys[1]).abs().max().item() / ys[1].abs().max().item()
    # (the two kernels round the normalised operand to fp16 once / twice -- a last-bit difference of single operand elements)
    assert d <= 2e-4, f"tap-folded and LDS-image kernels differ by {d:.2e} of the output range"


def test_conv_groupnorm_without_silu(lib):
    """GroupNorm prologue with silu_in = 0 (the instantiation without the activation) on the wide-tile kernel."""
    g = torch.Generator().manual_seed(3)
    B, C, H, W, Cout = 2, 128, 16, 32, 64
    x = torch.randn((B, C, H, W), generator=g)
    w = torch.randn((Cout, C, 3, 3), generator=g) / math.sqrt(9 * C)
    gn = (1.0 + 0.2 * torch.randn((B, C), generator=g), 0.2 * torch.randn((B, C), generator=g))
    got, ref = run_conv(lib, x, w, None, None, 1, (1, 1), 0, gn, 0, None, None, False, False)
    assert_close(got, ref, "gn_no_silu")


def test_conv_zero_padding_is_applied_after_groupnorm(lib):
    """The conv zero-pads the *normalised+activated* tensor: border taps must contribute exactly 0, not silu(shift)."""
    B, C, H, W = 1, 64, 4, 4
    x = torch.zeros((B, C, H, W))
    w = torch.ones((64, C, 3, 3)) / (9 * C)
    gn = (torch.ones((B, C)), torch.full((B, C), 2.0))  # silu(2.0) != 0 everywhere
    got, ref = run_conv(lib, x, w, None, None, 1, (1, 1), 0, gn, 1)
    assert_close(got, ref, "gn-pad")
    assert got[0, 0, 0, 0] < got[0, 0, 1, 1] * 0.6  # corner sees 4 of 9 taps


def test_conv_rejects_bad_arguments(lib):
    a_ = _lib.ConvArgs()
    x = torch.zeros(64, dtype=torch.float16, device=DEV)
    a_.x = a_.w = a_.y = x.data_ptr()
    a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout, a_.ks, a_.stride = 12, 1, 1, 1, 1, 1, 1, 1  # channels not % 8
    a_.N, a_.Nrows, a_.ldy = 16, 16, 16
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    a_.C1, a_.ks = 16, 5
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))


ATTN_CASES = {
    # name: (B, heads, Lq, Lk, d, kv_broadcast)
    "self_d40_L256": (2, 8, 256, 256, 40, False),
    "self_d40_L4096": (1, 8, 4096, 4096, 40, False),
    "self_d80_L1024": (2, 8, 1024, 1024, 80, False),
    "self_d160_L256": (2, 8, 256, 256, 160, False),
    "self_d160_L64": (2, 8, 64, 64, 160, False),
    "cross_d40_Lk6_bcast": (3, 8, 300, 6, 40, True),
    "cross_d80_Lk77": (2, 8, 128, 77, 80, False),
    "cross_d160_Lk1": (2, 8, 64, 1, 160, True),
    # short K / V (L_ctx <= 16): the all-heads-in-one-wave kernel (xattn_kernel); ragged query tiles, per-image K / V, the full 16 keys, other head counts
    "cross_d40_Lk6_bcast_big": (8, 8, 4096, 6, 40, True),
    "cross_d40_Lk16_per_image_ragged": (3, 8, 333, 16, 40, False),
    "cross_d80_Lk6_bcast": (4, 8, 1024, 6, 80, True),
    "cross_d80_Lk9_heads4": (2, 4, 100, 9, 80, False),
    "cross_d160_Lk6_bcast": (8, 8, 256, 6, 160, True),
    "cross_d160_Lk13_heads5": (1, 5, 64, 13, 160, False),
    "vae_d512_L1024": (2, 1, 1024, 1024, 512, False),
    "vae_d128_L256": (1, 1, 256, 256, 128, False),
    "tiny_d8": (2, 8, 256, 256, 8, False),
    "tiny_d16_ragged": (2, 8, 100, 37, 16, False),
    "tiny_d32": (2, 8, 64, 64, 32, False),
}


@pytest.mark.parametrize("name", list(ATTN_CASES))
def test_attention(lib, name):
    B, heads, Lq, Lk, d, bcast = ATTN_CASES[name]
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    Cc = heads * d
    q = torch.randn((B, Lq, Cc), generator=g)
    Bk = 1 if bcast else B
    k = torch.randn((Bk, Lk, Cc), generator=g)
    v = torch.randn((Bk, Lk, Cc), generator=g)
    qd, kd, vd = (t.to(torch.float16).to(DEV) for t in (q, k, v))
    o = torch.full((B, Lq, Cc), float("nan"), dtype=torch.float16, device=DEV)
    scale = 1.0 / math.sqrt(d)
    _lib.check(lib.ldiff_op_attention(qd.data_ptr(), Cc, kd.data_ptr(), Cc, vd.data_ptr(), Cc, o.data_ptr(), Cc, B, heads, Lq, Lk, d,
                                      Lq * Cc, 0 if bcast else Lk * Cc, Lq * Cc, scale, sp()))
    torch.cuda.synchronize()
    qh = r16(q).view(B, Lq, heads, d).transpose(1, 2)
    kh = r16(k).expand(B, -1, -1).reshape(B, Lk, heads, d).transpose(1, 2)
    vh = r16(v).expand(B, -1, -1).reshape(B, Lk, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, Lq, Cc)
    # probabilities are rounded to fp16 before P.V: allow 3e-3 of the value range
    assert_close(o, ref, name, rtol=3e-3, atol_rel=3e-3)


@pytest.mark.parametrize("d,heads,Lq,Lk", [(40, 8, 200, 300), (512, 1, 200, 130)])
def test_attention_long_shared_keys_on_the_fixed_reference_kernels(lib, d, heads, Lq, Lk):
    """K / V shared by every image of the batch (kv_bstride = 0) with more than one key tile: not a shape the samplers launch, but the two
    fixed-reference kernels accept it (their K / V buffer descriptors are built per image from the batch stride)."""
    B, Cc = 3, heads * d
    g = torch.Generator().manual_seed(d + Lq)
    q = torch.randn((B, Lq, Cc), generator=g)
    k = torch.randn((1, Lk, Cc), generator=g)
    v = torch.randn((1, Lk, Cc), generator=g)
    qd, kd, vd = (t.to(torch.float16).to(DEV) for t in (q, k, v))
    o = torch.full((B, Lq, Cc), float("nan"), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_attention(qd.data_ptr(), Cc, kd.data_ptr(), Cc, vd.data_ptr(), Cc, o.data_ptr(), Cc, B, heads, Lq, Lk, d,
                                      Lq * Cc, 0, Lq * Cc, 1.0 / math.sqrt(d), sp()))
    torch.cuda.synchronize()
    qh = r16(q).view(B, Lq, heads, d).transpose(1, 2)
    kh = r16(k).expand(B, -1, -1).reshape(B, Lk, heads, d).transpose(1, 2)
    vh = r16(v).expand(B, -1, -1).reshape(B, Lk, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, Lq, Cc)
    assert_close(o, ref, f"shared keys d={d}", rtol=3e-3, atol_rel=3e-3)


@pytest.mark.parametrize("name,B,heads,Lq,Lk,fused,late_spike", [
    ("level0", 2, 8, 1024, 1024, True, False),
    ("ragged_tails", 2, 8, 300, 333, False, False),
    ("other_head_count", 1, 5, 200, 130, False, False),
    ("repeat_pass", 2, 8, 384, 512, True, True),
    ("repeat_pass_two_clamped_keys", 2, 8, 384, 512, False, "pair"),
    ("repeat_pass_exp2_overflow", 1, 8, 256, 320, False, "huge"),
])
def test_attention_d40_fixed_reference(lib, name, B, heads, Lq, Lk, fused, late_spike):
    """attn_fr40_kernel (d = 40, at least two key tiles): fixed softmax reference per query (first key tile's maximum + 4 binades), row sums out of
    the P V MFMAs through a ones row, P packed round-toward-zero, K / V by LDS-DMA.  Cases: fused q/k/v rows (the UNet's layout), ragged query /
    key tails, a head count that is not 8, and late keys that overflow fp16 P against the first tile's reference (row sum inf -> the workgroup
    takes the true maxima in a scores-only pass and repeats)."""
    d = 40
    Cc = heads * d
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    q = torch.randn((B, Lq, Cc), generator=g)
    k = torch.randn((B, Lk, Cc), generator=g)
    v = torch.randn((B, Lk, Cc), generator=g)
    scale = 1.0 / math.sqrt(d)

    def key_at(bi, qi, h, binades):
        """a key along query (bi, qi) of head h whose score lies `binades` above that query's maximum over the first key tile (64 keys)"""
        qv = r16(q[bi, qi, h * d:(h + 1) * d])
        m1 = (r16(k[bi, :64, h * d:(h + 1) * d]) @ qv).max().item() * scale          # nats
        return qv * ((m1 + binades * math.log(2.0)) / scale / (qv @ qv).item())

    if late_spike is True:
        k[0, Lk - 3, 2 * d:3 * d] = 12.0 * q[0, 5, 2 * d:3 * d]        # head 2, query 5 of image 0: score 12 |q|^2 / sqrt(40) ~ 75 nats above the first tile
        k[1, Lk // 2 + 1, :d] = 8.0 * q[1, 290, :d]                    # another workgroup, a middle tile
    elif late_spike == "pair":
        # TWO late keys of one row, 30 and 27 binades above the first tile (both beyond fp16 P's window of 16 - 4 lead binades, both far inside
        # fp32 exp2's range): true weights 8 : 1.  Packed round-toward-zero both clamp to 65504 -- equal weights, a wrong softmax -- unless the
        # kernel notices the clamp and repeats the pass with the true maximum (the row-sum test must see a clamped P)
        k[0, Lk - 7, 3 * d:4 * d] = key_at(0, 9, 3, 30.0)
        k[0, Lk - 70, 3 * d:4 * d] = key_at(0, 9, 3, 27.0)
        k[1, 200, 6 * d:7 * d] = key_at(1, 300, 6, 17.5)              # just outside the window: one clamped key beside in-window mass
        k[1, 130, 6 * d:7 * d] = key_at(1, 300, 6, 14.0)
    elif late_spike == "huge":
        k[0, Lk - 1, :d] = key_at(0, 100, 0, 140.0)                    # exp2 overflows fp32: the row sum is inf / nan
        k[0, 90, :d] = key_at(0, 100, 0, 139.0)
    o = torch.full((B, Lq, Cc), float("nan"), dtype=torch.float16, device=DEV)
    if fused:
        L = max(Lq, Lk)
        buf = torch.zeros((B, L, 3 * Cc), dtype=torch.float16)
        buf[:, :Lq, :Cc] = q.to(torch.float16); buf[:, :Lk, Cc:2 * Cc] = k.to(torch.float16); buf[:, :Lk, 2 * Cc:] = v.to(torch.float16)
        buf = buf.to(DEV)
        base = buf.data_ptr()
        _lib.check(lib.ldiff_op_attention(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, Lq, Lk, d,
                                          L * 3 * Cc, L * 3 * Cc, Lq * Cc, scale, sp()))
    else:
        qd, kd, vd = (t.to(torch.float16).to(DEV) for t in (q, k, v))
        _lib.check(lib.ldiff_op_attention(qd.data_ptr(), Cc, kd.data_ptr(), Cc, vd.data_ptr(), Cc, o.data_ptr(), Cc, B, heads, Lq, Lk, d,
                                          Lq * Cc, Lk * Cc, Lq * Cc, scale, sp()))
    torch.cuda.synchronize()
    qh = r16(q).view(B, Lq, heads, d).transpose(1, 2)
    kh = r16(k).view(B, Lk, heads, d).transpose(1, 2)
    vh = r16(v).view(B, Lk, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, Lq, Cc)
    assert_close(o, ref, name, rtol=3e-3, atol_rel=3e-3)
    if late_spike == "pair":   # the case must be able to fail: equal weights on the two clamped keys are far outside the tolerance
        wrong = 0.5 * (r16(v[0, Lk - 7, 3 * d:4 * d]) + r16(v[0, Lk - 70, 3 * d:4 * d]))
        assert (wrong - ref[0, 9, 3 * d:4 * d]).abs().max() > 0.1


@pytest.mark.parametrize("name,B,Lq,Lk,fused,late_spike", [
    ("mid_block_1024", 2, 1024, 1024, True, False),
    ("ragged_tails", 3, 200, 333, False, False),
    ("one_query_block_short_keys", 1, 128, 17, False, False),
    ("second_pass", 2, 384, 512, True, True),
])
def test_attention_d512_fixed_reference(lib, name, B, Lq, Lk, fused, late_spike):
    """attn_d512_kernel (d = 512, one head, more than 64 queries): 128 queries per workgroup, K / V by LDS-DMA, and a FIXED softmax reference per
    query (maximum over the first key tile + 4 binades) instead of a running one -- the accumulators are never rescaled.  Cases: the fused q/k/v
    layout of the VAE mid block, ragged query / key tails, fewer keys than one tile, and keys in a late tile whose scores leave the fp16 window of
    the first tile's reference (the workgroup then repeats the pass with the true maxima)."""
    d = 512
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    q = torch.randn((B, Lq, d), generator=g)
    k = torch.randn((B, Lk, d), generator=g)
    v = torch.randn((B, Lk, d), generator=g)
    if late_spike:
        k[0, Lk - 5] = 2.0 * q[0, 7]          # score 2 |q|^2 / sqrt(512) ~ 45 nats above everything in the first tile: query 7 of image 0 leaves the window
        k[1, Lk // 2] = 1.2 * q[1, 300]       # ~ 27 nats: another workgroup, another tile
    scale = 1.0 / math.sqrt(d)
    o = torch.full((B, Lq, d), float("nan"), dtype=torch.float16, device=DEV)
    if fused:
        assert Lq == Lk or True
        L = max(Lq, Lk)
        buf = torch.zeros((B, L, 3 * d), dtype=torch.float16)
        buf[:, :Lq, :d] = q.to(torch.float16); buf[:, :Lk, d:2 * d] = k.to(torch.float16); buf[:, :Lk, 2 * d:] = v.to(torch.float16)
        buf = buf.to(DEV)
        base = buf.data_ptr()
        _lib.check(lib.ldiff_op_attention(base, 3 * d, base + 2 * d, 3 * d, base + 4 * d, 3 * d, o.data_ptr(), d, B, 1, Lq, Lk, d,
                                          L * 3 * d, L * 3 * d, Lq * d, scale, sp()))
    else:
        qd, kd, vd = (t.to(torch.float16).to(DEV) for t in (q, k, v))
        _lib.check(lib.ldiff_op_attention(qd.data_ptr(), d, kd.data_ptr(), d, vd.data_ptr(), d, o.data_ptr(), d, B, 1, Lq, Lk, d,
                                          Lq * d, Lk * d, Lq * d, scale, sp()))
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(r16(q)[:, None], r16(k)[:, None], r16(v)[:, None])[:, 0]
    assert_close(o, ref, name, rtol=3e-3, atol_rel=3e-3)


@pytest.mark.parametrize("B,heads,L,d,spike", [(2, 8, 320, 40, True), (1, 8, 4096, 40, False), (2, 8, 1024, 80, True), (1, 8, 100, 40, False), (1, 4, 64, 80, False)])
def test_attention_prescaled(lib, B, heads, L, d, spike):
    """ldiff_op_attention_prescaled: q arrives multiplied by scale * log2(e) and rounded once (as the q/k/v projection's epilogue delivers it); the
    kernel subtracts the running reference inside the MFMAs.  Against SDPA on the same (rounded) operands: the reference softmax(q' k / log2 e)
    v; a spiky key in a late tile forces the fix-up branch (new reference, rescale of O, update of the Q fragment), an all-negative first tile
    the negative-reference start; ragged tails; fused q/k/v layout."""
    Cc = heads * d
    g = torch.Generator().manual_seed(L + d)
    qkv = torch.randn((B, L, 3 * Cc), generator=g)
    if spike:
        qkv[0, L - 20, Cc:2 * Cc] *= 12.0                                   # a key in the last tile that moves many rows' maxima
        qkv[-1, :64, Cc:2 * Cc] = -qkv[-1, :64, :Cc].mean(0, keepdim=True) * 0 - 3.0 * torch.sign(qkv[-1, :1, :Cc])   # first tile: scores mostly negative
    sl2 = (1.0 / math.sqrt(d)) * 1.4426950408889634
    qs = (qkv[..., :Cc] * sl2)                                             # the producer's fp32 value, rounded once below
    buf = torch.cat([qs, qkv[..., Cc:]], -1).to(torch.float16).to(DEV)
    o = torch.full((B, L, Cc), float("nan"), dtype=torch.float16, device=DEV)
    base = buf.data_ptr()
    _lib.check(lib.ldiff_op_attention_prescaled(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, L, L, d,
                                                L * 3 * Cc, L * 3 * Cc, L * Cc, sp()))
    torch.cuda.synchronize()
    x = buf.float().cpu()
    sh = lambda t: t.view(B, L, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sh(x[..., :Cc]), sh(x[..., Cc:2 * Cc]), sh(x[..., 2 * Cc:]), scale=1.0 / 1.4426950408889634).transpose(1, 2).reshape(B, L, Cc)
    assert_close(o, ref, f"prescaled attention B={B} L={L} d={d}", rtol=3e-3, atol_rel=3e-3)
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_attention_prescaled(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads * d // 160 or 1, L, L, 160,
                                                    L * 3 * Cc, L * 3 * Cc, L * Cc, sp()))


def test_attention_fused_qkv_layout_and_online_softmax_spike(lib):
    """q/k/v interleaved in one [B, L, 3C] buffer (the self-attention call) + a key that forces a large running-max jump
    in a late tile (exercises the rescale branch of the online softmax)."""
    B, heads, L, d = 1, 8, 320, 40
    Cc = heads * d
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn((B, L, 3 * Cc), generator=g)
    qkv[0, 300, Cc:2 * Cc] *= 12.0  # spiky key in the last tile
    dqkv = qkv.to(torch.float16).to(DEV)
    o = torch.empty((B, L, Cc), dtype=torch.float16, device=DEV)
    base = dqkv.data_ptr()
    _lib.check(lib.ldiff_op_attention(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, L, L, d,
                                      L * 3 * Cc, L * 3 * Cc, L * Cc, 1.0 / math.sqrt(d), sp()))
    torch.cuda.synchronize()
    x = r16(qkv)
    sh = lambda t: t.view(B, L, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sh(x[..., :Cc]), sh(x[..., Cc:2 * Cc]), sh(x[..., 2 * Cc:])).transpose(1, 2).reshape(B, L, Cc)
    assert_close(o, ref, "fused-qkv-spike", rtol=3e-3, atol_rel=3e-3)


GN_CASES = {
    # name: (B, C1, C2, HW, groups, eps)
    "c320_hw4096": (2, 320, 0, 4096, 32, 1e-5),
    "c128_hw16384_eps6": (2, 128, 0, 16384, 32, 1e-6),
    "concat_1280_640_straddle": (2, 1280, 640, 64, 32, 1e-5),
    "concat_128_64": (3, 128, 64, 256, 32, 1e-5),
    "c64_cg2": (2, 64, 0, 300, 32, 1e-5),
    "c32_cg1": (2, 32, 0, 1024, 32, 1e-6),
    "hw1": (2, 64, 0, 1, 32, 1e-5),
    "large_mean": (2, 64, 0, 4096, 32, 1e-5),
    # one-launch small-map kernel (HW <= 1024, B * groups >= 64): groups of 20 / 60 channels cut through 16-byte chunks
    "small_c640_cg20_hw1024": (4, 640, 0, 1024, 32, 1e-5),
    "small_concat_1280_640_hw256": (8, 1280, 640, 256, 32, 1e-6),
    "small_c2560_hw64": (8, 1280, 1280, 64, 32, 1e-5),
}


@pytest.mark.parametrize("name", list(GN_CASES))
def test_group_norm_stats(lib, name):
    B, C1, C2, HW, groups, eps = GN_CASES[name]
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    Cc = C1 + C2
    x = torch.randn((B, HW, Cc), generator=g) * 1.7 + 0.3
    if name == "large_mean":
        x = x * 0.1 + 30.0  # |mean|/std = 300: the E[x^2]-mean^2 cancellation case
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    x16 = x.to(torch.float16)
    x1 = x16[..., :C1].contiguous().to(DEV)
    x2 = x16[..., C1:].contiguous().to(DEV) if C2 else None
    scale = torch.empty((B, Cc), device=DEV)
    shift = torch.empty((B, Cc), device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    _lib.check(lib.ldiff_op_gn_stats(x1.data_ptr(), C1, 0, 0, x2.data_ptr() if C2 else None, C2, 0, 0, B, HW, groups, eps, gd.data_ptr(), bd.data_ptr(),
                                     scale.data_ptr(), shift.data_ptr(), sp()))
    torch.cuda.synchronize()
    xr = x16.float()
    got = xr * scale.cpu()[:, None, :] + shift.cpu()[:, None, :]
    ref = F.group_norm(xr.permute(0, 2, 1), groups, gamma, beta, eps).permute(0, 2, 1)
    tol = 2e-2 if name == "large_mean" else 2e-4
    assert (got - ref).abs().max() <= tol * max(1.0, ref.abs().max()), f"{name}: max err {(got - ref).abs().max():.3e}"


@pytest.mark.parametrize("rows,C", [(4096, 320), (1024, 640), (513, 1280), (7, 64), (1, 2560)])
def test_layernorm(lib, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn((rows, C), generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    xd, gd, bd = x.to(torch.float16).to(DEV), gamma.to(DEV), beta.to(DEV)
    y = torch.empty_like(xd)
    _lib.check(lib.ldiff_op_layernorm(xd.data_ptr(), 0, 0, y.data_ptr(), rows, C, gd.data_ptr(), bd.data_ptr(), 1e-5, sp()))
    torch.cuda.synchronize()
    ref = F.layer_norm(r16(x), (C,), gamma, beta, 1e-5)
    assert_close(y, ref, f"layernorm {rows}x{C}", rtol=1e-3, atol_rel=1e-3)


@pytest.mark.parametrize("M,C4", [(4096, 1280), (300, 2560), (1, 256)])
def test_geglu(lib, M, C4):
    g = torch.Generator().manual_seed(M)
    x = torch.randn((M, 2 * C4), generator=g) * 1.5
    xd = x.to(torch.float16).to(DEV)
    y = torch.empty((M, C4), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_geglu(xd.data_ptr(), y.data_ptr(), M, C4, sp()))
    torch.cuda.synchronize()
    xr = r16(x)
    ref = xr[:, :C4] * F.gelu(xr[:, C4:])
    assert_close(y, ref, "geglu", rtol=1e-3, atol_rel=1e-3)


def test_layout_nchw_to_nhwc_pads_channels(lib):
    x = torch.randn((2, 3, 5, 7))
    xd = x.to(DEV)
    y = torch.full((2, 5, 7, 8), float("nan"), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_nchw_to_nhwc(xd.data_ptr(), y.data_ptr(), 2, 3, 5, 7, 8, 0, sp()))
    torch.cuda.synchronize()
    assert torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 1).to(torch.float16))
    assert (y[..., 3:] == 0).all()
    # split form: hi at channels 0..2, the rounding remainder at 3..5, zeros after
    _lib.check(lib.ldiff_op_nchw_to_nhwc(xd.data_ptr(), y.data_ptr(), 2, 3, 5, 7, 8, 3, sp()))
    torch.cuda.synchronize()
    xh = x.permute(0, 2, 3, 1)
    hi = xh.to(torch.float16)
    assert torch.equal(y[..., :3].cpu(), hi) and torch.equal(y[..., 3:6].cpu(), (xh - hi.float()).to(torch.float16)) and (y[..., 6:] == 0).all()
    assert (y[..., :3].float().cpu() + y[..., 3:6].float().cpu() - xh).abs().max() <= 2e-6 * xh.abs().max()
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_nchw_to_nhwc(xd.data_ptr(), y.data_ptr(), 2, 3, 5, 7, 8, 6, sp()))   # lo half does not fit


FUSED_STATS_CASES = {
    # name: (B, C1, H, W, Cout, ks, stride)   -- producer conv emits GroupNorm partial sums in its epilogue
    "conv3x3_halo_128": (2, 64, 32, 32, 128, 3, 1),
    "conv3x3_halo_partial_tile_320": (2, 64, 20, 24, 320, 3, 1),
    "conv3x3_small_image_8x8": (3, 128, 8, 8, 128, 3, 1),
    "gemm_dma_1x1": (2, 128, 16, 16, 256, 1, 1),
    "igemm_stride2": (2, 64, 32, 32, 64, 3, 2),
    "conv3x3_x16_tile": (8, 64, 128, 128, 128, 3, 1),
    "conv3x3_p16_one_tile_per_workgroup": (1, 64, 256, 256, 128, 3, 1),
    "conv3x3_p16_ragged_72x88": (8, 64, 72, 88, 128, 3, 1),
    # round 6: split-K launches -- the statistics come from the reduce kernel (32-row blocks), whichever kernel wrote the partials
    "conv3x3_8x8_split4": (3, 256, 8, 8, 128, 3, 1, 4),
    "conv3x3_8x8_split16_of_20_slabs": (1, 1280, 8, 8, 128, 3, 1, 16),
    "conv3x3_16x16_split2_320": (2, 128, 16, 16, 320, 3, 1, 2),
    # the 160-column kernel's three-slot weight ring: K ranges that start at a later slab and run over more than one (the next slab's first two slices
    # are fetched during the last two taps of the current one)
    "conv3x3_16x16_split2_320_two_slabs_each": (1, 256, 16, 16, 320, 3, 1, 2),
    "conv3x3_32x32_split3_320_of_7_slabs": (1, 448, 32, 32, 320, 3, 1, 3),
    "gemm_dma_1x1_split3": (2, 512, 16, 16, 256, 1, 1, 3),
    "gemm_dma_1x1_split9_tail_rows": (1, 1280, 8, 8, 320, 1, 1, 9),
    "igemm_stride2_split2": (2, 64, 32, 32, 64, 3, 2, 2),
}


@pytest.mark.parametrize("name", list(FUSED_STATS_CASES))
def test_fused_groupnorm_statistics(lib, name):
    """Statistics accumulated in the producer's epilogue + finalize == F.group_norm of the (fp16) output it wrote."""
    B, C1, H, W, Cout, ks, stride = FUSED_STATS_CASES[name][:7]
    splitk = FUSED_STATS_CASES[name][7] if len(FUSED_STATS_CASES[name]) > 7 else 0
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    x = torch.randn((B, C1, H, W), generator=g)
    w = torch.randn((Cout, C1, ks, ks), generator=g) / math.sqrt(C1 * ks * ks)
    bias = torch.randn(Cout, generator=g) * 0.5
    Ho, Wo = (H + 2 * (ks // 2) - ks) // stride + 1, (W + 2 * (ks // 2) - ks) // stride + 1
    wd = w.permute(0, 2, 3, 1).reshape(Cout, -1).to(torch.float16).contiguous().to(DEV)
    xd, bd = nhwc16(x), bias.to(DEV)
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device=DEV)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = xd.data_ptr(), C1, B, H, W, Ho, Wo
    a_.ks, a_.stride, a_.pad_t, a_.pad_l = ks, stride, ks // 2, ks // 2
    a_.w, a_.N, a_.Nrows, a_.bias, a_.y, a_.ldy = wd.data_ptr(), Cout, Cout, bd.data_ptr(), y.data_ptr(), Cout
    if splitk:   # the unsplit launch first: the split one must reproduce its output (fp32 partial sums in another association)
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        y_unsplit = y.clone()
        y.fill_(float("nan"))
        a_.splitk = splitk
    R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
    assert R > 0 and (not splitk or R == Ho * Wo // 32)
    st = torch.full((B, Cout, R, 2), float("nan"), device=DEV)
    a_.stats = st.data_ptr()
    _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    gamma, beta = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    scale, shift = torch.empty((B, Cout), device=DEV), torch.empty((B, Cout), device=DEV)
    gd, btd = gamma.to(DEV), beta.to(DEV)
    _lib.check(lib.ldiff_op_gn_finalize(st.data_ptr(), R, Cout, None, 0, 0, B, Ho * Wo, 32, 1e-5, gd.data_ptr(), btd.data_ptr(),
                                        scale.data_ptr(), shift.data_ptr(), sp()))
    torch.cuda.synchronize()
    assert torch.isfinite(st).all()
    if splitk:
        assert torch.isfinite(y).all() and (y.float() - y_unsplit.float()).abs().max() <= 2e-3 * y_unsplit.float().abs().max()
    yr = y.float().cpu().permute(0, 3, 1, 2)
    got = yr * scale.cpu()[:, :, None, None] + shift.cpu()[:, :, None, None]
    ref = F.group_norm(yr, 32, gamma, beta, 1e-5)
    assert (got - ref).abs().max() <= 2e-4 * max(1.0, ref.abs().max())


@pytest.mark.parametrize("M,Cc,inner", [(300, 64, 128), (4096, 320, 1280), (513, 128, 512)])
def test_linear_with_fused_geglu_epilogue(lib, M, Cc, inner):
    """diffusers GEGLU (Linear(C, 2*inner) -> x * gelu_erf(gate)) as the GEMM epilogue: weight rows interleaved by 16 at load."""
    g = torch.Generator().manual_seed(M + Cc)
    x = torch.randn((M, Cc), generator=g)
    w = torch.randn((2 * inner, Cc), generator=g) / math.sqrt(Cc)
    b = torch.randn(2 * inner, generator=g) * 0.2
    proj = r16(x) @ r16(w).t() + b
    ref = proj[:, :inner] * F.gelu(proj[:, inner:])
    perm = torch.empty(2 * inner, dtype=torch.long)
    for r in range(2 * inner):
        q = r if r < inner else r - inner
        perm[(q // 16) * 32 + (0 if r < inner else 16) + q % 16] = r
    wd = w[perm].to(torch.float16).contiguous().to(DEV)
    bd = b[perm].contiguous().to(DEV)
    xd = x.to(torch.float16).contiguous().to(DEV)
    y = torch.full((M, inner), float("nan"), dtype=torch.float16, device=DEV)
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride = xd.data_ptr(), Cc, 1, 1, M, 1, M, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy, a.geglu = wd.data_ptr(), 2 * inner, 2 * inner, bd.data_ptr(), y.data_ptr(), inner, 1
    _lib.check(lib.ldiff_op_conv(C.byref(a), sp()))
    torch.cuda.synchronize()
    assert_close(y.float().cpu(), ref, f"geglu_{M}_{Cc}_{inner}")
    a.res, a.ld_res = y.data_ptr(), inner                       # a residual cannot be combined with the GEGLU epilogue
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp()))


# ======================================================================================================================
# Producer / consumer ("dataflow") GEMM (kernels_gemm_df.hip), forced through ldiff_conv_args.gemm_df = 16 mt + ntw
# ======================================================================================================================
GEMM_DF_CASES = {
    # name: (M, K1, K2, N, epilogue, plan 16 mt + ntw or 1 = the launcher's own plan)
    "plain_one_unit_per_workgroup": (2048, 320, 0, 320, "plain", 16 * 8 + 5),
    "plain_bn128_tail_columns": (1024, 320, 0, 320, "plain", 16 * 8 + 2),            # N = 2.5 units of 128 columns
    "plain_runs_of_units": (300 * 128, 64, 0, 640, "plain", 16 * 8 + 5),              # 600 units on <= 256 workgroups: runs of 2-3 units, K = one step
    "plain_tail_rows": (1000, 128, 0, 256, "plain", 16 * 8 + 4),                      # M = 7.8 row blocks
    "plain_mt4": (520, 192, 0, 384, "plain", 16 * 4 + 2),
    "plain_mt4_ntw5": (8192, 640, 0, 640, "plain", 16 * 4 + 5),
    "plain_mt4_ntw4_no_bias": (4096, 1280, 0, 1280, "plain_nobias", 16 * 4 + 4),
    "res_plain": (4096, 320, 0, 320, "res", 16 * 8 + 5),
    "split_out": (4096, 640, 0, 320, "split_out", 16 * 8 + 5),
    "split_res_split_out": (8192, 320, 0, 320, "split_res_out", 16 * 8 + 5),
    "split_res_split_out_long_k": (4096, 1280, 0, 320, "split_res_out", 16 * 8 + 2),
    "split_res_split_out_mt4_tails": (1000, 256, 0, 200, "split_res_out", 16 * 4 + 2),
    "split_res_plain_out": (2048, 320, 0, 640, "split_res", 16 * 8 + 4),
    "concat_two_sources": (2048, 128, 64, 320, "split_out", 16 * 8 + 5),              # the 1x1 shortcut over [x | skip]
    "concat_two_sources_pitch": (2048, 192, 320, 256, "plain", 16 * 8 + 2),
    "geglu": (4096, 320, 0, 2560, "geglu", 16 * 8 + 5),
    "geglu_bn128_tail_rows": (1000, 128, 0, 768, "geglu", 16 * 8 + 2),
    "geglu_mt4_ntw4": (2048, 640, 0, 5120, "geglu", 16 * 4 + 4),
    "stats_split_res_split_out": (8192, 640, 0, 320, "split_res_out+stats", 16 * 8 + 5),     # proj_out of Transformer2DModel at level 0: two images of 4096 rows
    "stats_plain_res_mt4": (2048, 256, 0, 256, "res+stats", 16 * 4 + 4),
    "stats_split_out_bn128_tail_columns": (4096, 128, 0, 320, "split_out+stats", 16 * 8 + 2),
    "auto_plan_level1_ff2": (8192, 2560, 0, 640, "split_res_out", 1),
    "auto_plan_level2_qkv": (2048, 1280, 0, 3840, "plain_nobias", 1),
}


GEMM_DF_SPLITK_ON_DMA = set()   # cases where gemm_dma_splitk_plan > 1 (filled if a case is added that has one; none of the above does)


@pytest.mark.parametrize("name", list(GEMM_DF_CASES))
def test_gemm_dataflow(lib, name):
    """gemm_df_kernel against a float64 GEMM on the operands the kernel sees: every epilogue (bias, plain / split residual, plain / split output, GEGLU),
    every unit shape, ragged rows and columns, two concat sources, runs of several units per workgroup.  Plain outputs: one fp16 rounding of the fp32
    result; split outputs: hi + lo to fp32 round-off."""
    M, K1, K2, N, epi, plan = GEMM_DF_CASES[name]
    want_stats = epi.endswith("+stats")
    epi = epi.replace("+stats", "")
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    K = K1 + K2
    geglu = epi == "geglu"
    x1 = torch.randn((M, K1), generator=g)
    x2 = torch.randn((M, K2), generator=g) if K2 else None
    w = r16(torch.randn((N, K), generator=g) / math.sqrt(K))
    bias = None if epi == "plain_nobias" else torch.randn(N, generator=g) * 0.3
    Nrows = (N + 15) // 16 * 16
    a_ = _lib.ConvArgs()
    keep = []
    if name == "concat_two_sources_pitch":      # rows with a pitch: the hi halves of split tensors
        x1d, x2d = to_split(x1).to(DEV), to_split(x2).to(DEV)
        a_.ld1, a_.ld2 = 2 * K1, 2 * K2
    else:
        x1d, x2d = x1.to(torch.float16).to(DEV), (x2.to(torch.float16).to(DEV) if K2 else None)
    a_.x, a_.C1, a_.C2 = x1d.data_ptr(), K1, K2
    if K2:
        a_.x2 = x2d.data_ptr()
    A = torch.cat([r16(x1)] + ([r16(x2)] if K2 else []), 1).double()
    if geglu:
        inner = N // 2
        perm = torch.empty(N, dtype=torch.long)
        for r in range(N):
            q = r if r < inner else r - inner
            perm[(q // 16) * 32 + (0 if r < inner else 16) + q % 16] = r
        wdev, bdev = w[perm], bias[perm]
    else:
        wdev, bdev = w, bias
    wd = torch.zeros((Nrows, K), dtype=torch.float16); wd[:N] = wdev.to(torch.float16); wd = wd.to(DEV)
    a_.w, a_.N, a_.Nrows = wd.data_ptr(), N, Nrows
    if bdev is not None:
        bd = torch.zeros(Nrows); bd[:N] = bdev; bd = bd.to(DEV); keep.append(bd)
        a_.bias = bd.data_ptr()
    Bimg = 2 if want_stats else 1          # statistics are per image: two images of M / 2 rows
    a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout, a_.ks, a_.stride = Bimg, 1, M // Bimg, 1, M // Bimg, 1, 1
    ref = A @ w.double().t() + (bias.double() if bias is not None else 0.0)
    split_out = epi in ("split_out", "split_res_out")
    Nout = N // 2 if geglu else N
    if geglu:
        ref = ref[:, :Nout] * F.gelu(ref[:, Nout:])
    if epi in ("res", "split_res", "split_res_out"):
        res = torch.randn((M, N), generator=g) * 3.0
        if epi == "res":
            rd = res.to(torch.float16).to(DEV)
            a_.res, a_.ld_res = rd.data_ptr(), N
            ref = ref + r16(res).double()
        else:
            rs = to_split(res); rd = rs.to(DEV)
            a_.res, a_.ld_res, a_.res_lo = rd.data_ptr(), 2 * N, N
            ref = ref + from_split(rs, N).double()
        keep.append(rd)
    y = torch.full((M, 2 * Nout if split_out else Nout), float("nan"), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy, a_.y_lo, a_.geglu = y.data_ptr(), y.shape[1], Nout if split_out else 0, int(geglu)
    a_.gemm_df = plan
    if want_stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
        assert R == M // Bimg // 32
        st = torch.full((Bimg, N, R, 2), float("nan"), device=DEV)
        a_.stats = st.data_ptr()
    _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    torch.cuda.synchronize()
    yc = y.cpu()
    assert torch.isfinite(yc.float()).all(), "unwritten or non-finite outputs"
    if want_stats:      # partial sums of the stored values (fp32 value of a split output, rounded value of a plain one) per 32-row block and channel
        stored = (from_split(yc, Nout) if split_out else yc.float()).double().view(Bimg, R, 32, N)
        sref = torch.stack([stored.sum(2), (stored * stored).sum(2)], -1).permute(0, 2, 1, 3)      # [B, N, R, 2]
        stc = st.cpu().double()
        assert torch.isfinite(stc).all(), "unwritten statistics"
        assert (stc - sref).abs().max() <= 2e-5 * sref.abs().max()
    if split_out:
        got = from_split(yc, Nout).double()
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        print(f"{name}: split output rel err {err:.2e}")
        assert err <= 1e-5
    else:
        assert_close(yc.float(), ref.float(), name)
    # the same launch on the LDS-DMA GEMM (gemm_df = -1): the two kernels add in the same order (K ascending in 32-deep MFMAs, then bias, residual hi,
    # residual lo) and must agree BIT FOR BIT -- the executors send a layer to one or the other by its row count, and a tile sampled alone must equal
    # the same tile inside a batch (test_tiles_are_independent_units).  gemm_dma's split-K plans sum K in another order: those shapes are left out.
    y2 = torch.full_like(y, float("nan"))
    a_.y, a_.gemm_df = y2.data_ptr(), -1
    if want_stats:
        st2 = torch.full_like(st, float("nan"))
        a_.stats = st2.data_ptr()
    _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    torch.cuda.synchronize()
    if name not in GEMM_DF_SPLITK_ON_DMA:
        assert torch.equal(y2, y), f"dataflow and LDS-DMA GEMM differ in {int((y2 != y).sum())} elements"
        if want_stats:
            assert torch.equal(st2, st), f"fused statistics differ in {int((st2 != st).sum())} of {st.numel()} sums (max {float((st2 - st).abs().max()):.3e})"


def test_gemm_dataflow_rejects_what_it_does_not_take(lib):
    a_ = _lib.ConvArgs()
    x = torch.zeros((256, 96), dtype=torch.float16, device=DEV)
    w = torch.zeros((64, 96), dtype=torch.float16, device=DEV)
    y = torch.zeros((256, 64), dtype=torch.float16, device=DEV)
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout, a_.ks, a_.stride = x.data_ptr(), 96, 1, 1, 256, 1, 256, 1, 1
    a_.w, a_.N, a_.Nrows, a_.y, a_.ldy, a_.gemm_df = w.data_ptr(), 64, 64, y.data_ptr(), 64, 1
    with pytest.raises(ValueError):      # K % 64 != 0
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    x = torch.zeros((256, 128), dtype=torch.float16, device=DEV)
    w = torch.zeros((64, 128), dtype=torch.float16, device=DEV)
    a_.x, a_.C1, a_.w, a_.gemm_df = x.data_ptr(), 128, w.data_ptr(), 16 * 8 + 3
    with pytest.raises(ValueError):      # a unit shape that is not built
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))


# ======================================================================================================================
# Split tensors (fp16 hi | lo per row, value = hi + lo): the residual stream of the UNet / VAE
# ======================================================================================================================
def to_split(x_nhwc):
    """[..., C] f32 -> [..., 2C] f16 = [hi | lo]"""
    hi = x_nhwc.to(torch.float16)
    lo = (x_nhwc - hi.float()).to(torch.float16)
    return torch.cat([hi, lo], -1).contiguous()


def from_split(t, C):
    return t[..., :C].float() + t[..., C:2 * C].float()


SPLIT_CASES = {
    # name: (B, C1, C2, H, W, Cout, ks, stride, ups, operand_split, gn)
    "wide3x3_hi_operand_gn": (2, 64, 0, 16, 32, 128, 3, 1, 0, False, True),          # resnet conv1 reading a split x through the GN prologue
    "wide3x3_hi_operand_concat_gn": (1, 128, 64, 16, 16, 64, 3, 1, 0, False, True),   # ... with a split skip tensor
    "small3x3_8x8_hi_operand": (2, 128, 0, 8, 8, 128, 3, 1, 0, False, True),
    "splitk3x3_hi_operand": (1, 1280, 0, 8, 8, 128, 3, 1, 0, False, True),
    "gemm_split_operand_shortcut_concat": (2, 128, 64, 16, 16, 128, 1, 1, 0, True, False),   # conv_shortcut over [x | skip], both split
    "gemm_split_operand_proj": (1, 320, 0, 1, 300, 320, 1, 1, 0, True, False),
    "gemm_split_operand_shortcut_splitk": (8, 640, 640, 8, 8, 256, 1, 1, 0, True, False),    # K = 2 x 1280, 32 tiles -> split-K
    "igemm_stride2_split_operand": (2, 64, 0, 16, 16, 64, 3, 2, 0, True, False),       # downsampler on the split stream
    "igemm_stride2_split_operand_splitk": (2, 320, 0, 16, 16, 192, 3, 2, 0, True, False),   # ... deep level: few tiles, K = 2 x 2880 -> split-K
    "wide3x3_upsample_split_operand": (1, 64, 0, 16, 16, 64, 3, 1, 1, True, False),    # upsampler (parity folding on duplicated weights)
    "wide3x3_split_operand": (1, 64, 0, 16, 16, 96, 3, 1, 0, True, False),             # PREC_FULL conv on a normalised split operand
    "x16_hi_operand_gn": (8, 128, 0, 128, 128, 128, 3, 1, 0, False, True),             # large map, two slabs, split tensors
    "x16_split_operand_upsample": (8, 64, 0, 64, 64, 64, 3, 1, 1, True, False),
    # persistent 16x16-tile kernel with every epilogue option at once (split residual, split output, fused statistics)
    "p16_split_operand_res_lo_stats": (8, 64, 0, 128, 128, 128, 3, 1, 0, True, False),
    "p16_hi_operand_res_lo_stats": (8, 128, 0, 128, 128, 128, 3, 1, 0, False, False),
    "p16_split_operand_one_tile_per_workgroup": (1, 64, 0, 256, 256, 128, 3, 1, 0, True, False),
    "p16_split_operand_ragged_40x100": (8, 64, 0, 40, 100, 128, 3, 1, 0, True, False),
}


@pytest.mark.parametrize("name", list(SPLIT_CASES))
def test_conv_on_split_tensors(lib, name):
    """Sources, residual and output as split tensors.  With a hi-only operand the MFMA sees fp16(x); with a split operand it sees
    x to ~22 bits.  Either way the residual add and the stored output must be exact to fp32 round-off: |hi+lo - ref| <= 3e-6."""
    B, C1, C2, H, W, Cout, ks, stride, ups, opsplit, use_gn = SPLIT_CASES[name]
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    Cin = C1 + C2
    x = torch.randn((B, H, W, C1), generator=g)
    x2 = torch.randn((B, H, W, C2), generator=g) if C2 else None
    w = r16(torch.randn((Cout, Cin, ks, ks), generator=g) / math.sqrt(Cin * ks * ks))
    bias = torch.randn(Cout, generator=g) * 0.1
    gn = (1.0 + 0.2 * torch.randn((B, Cin), generator=g), 0.2 * torch.randn((B, Cin), generator=g)) if use_gn else None
    He, We = H << ups, W << ups
    Ho, Wo = (He + 2 * (ks // 2) - ks) // stride + 1, (We + 2 * (ks // 2) - ks) // stride + 1
    res = torch.randn((B, Ho, Wo, Cout), generator=g) * 3.0
    xs, x2s = to_split(x), (to_split(x2) if C2 else None)
    # ---- reference (float64): the operand the kernel sees, exact weights, exact residual hi + lo ----
    seen = (lambda t, C: from_split(t, C)) if opsplit else (lambda t, C: t[..., :C].float())
    a = seen(xs, C1) if not C2 else torch.cat([seen(xs, C1), seen(x2s, C2)], -1)
    a = a.permute(0, 3, 1, 2).double()
    if gn is not None:
        a = F.silu(a * gn[0].double()[:, :, None, None] + gn[1].double()[:, :, None, None])
        a = r16(a.float()).double()
    if ups:
        a = F.interpolate(a, scale_factor=2.0, mode="nearest")
    rs = to_split(res)
    ref = F.conv2d(a, w.double(), bias.double(), stride=stride, padding=ks // 2) + from_split(rs, Cout).permute(0, 3, 1, 2).double()
    # ---- device ----
    Nrows = (Cout + 15) // 16 * 16
    wd = torch.zeros((Nrows, ks, ks, Cin), dtype=torch.float16)
    wd[:Cout] = w.permute(0, 2, 3, 1).to(torch.float16)
    wd = wd.reshape(Nrows, ks * ks * Cin).contiguous().to(DEV)
    a_ = _lib.ConvArgs()
    xd = xs.to(DEV)
    x2d = x2s.to(DEV) if C2 else None
    if opsplit:
        wdup = torch.full((Nrows, ks * ks * 2 * Cin), float("nan"), dtype=torch.float16, device=DEV)
        _lib.check(lib.ldiff_op_dup_weights(wd.data_ptr(), wdup.data_ptr(), Nrows, ks * ks, Cin, C1, C2, 2 * Cin, sp()))
        a_.w, a_.C1, a_.C2 = wdup.data_ptr(), 2 * C1, 2 * C2
    else:
        a_.w, a_.C1, a_.C2, a_.ld1, a_.ld2 = wd.data_ptr(), C1, C2, 2 * C1, 2 * C2
    a_.x = xd.data_ptr()
    if C2:
        a_.x2 = x2d.data_ptr()
    a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = B, H, W, Ho, Wo
    a_.ks, a_.stride, a_.ups, a_.pad_t, a_.pad_l = ks, stride, ups, ks // 2, ks // 2
    a_.N, a_.Nrows = Cout, Nrows
    if gn is not None:
        sd, hd = gn[0].contiguous().to(DEV), gn[1].contiguous().to(DEV)
        a_.gn_scale, a_.gn_shift, a_.silu_in = sd.data_ptr(), hd.data_ptr(), 1
    bd = torch.zeros(Nrows); bd[:Cout] = bias; bd = bd.to(DEV)
    a_.bias = bd.data_ptr()
    rd = rs.to(DEV)
    a_.res, a_.ld_res, a_.res_lo = rd.data_ptr(), 2 * Cout, Cout
    y = torch.full((B, Ho, Wo, 2 * Cout), float("nan"), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy, a_.y_lo = y.data_ptr(), 2 * Cout, Cout
    R = lib.ldiff_op_conv_stats_blocks(C.byref(a_)) if name != "splitk3x3_hi_operand" else 0
    if R > 0:
        st = torch.full((B, Cout, R, 2), float("nan"), device=DEV)
        a_.stats = st.data_ptr()
    _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    torch.cuda.synchronize()
    yc = y.cpu()
    got = from_split(yc, Cout).permute(0, 3, 1, 2).double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name}: split output rel err {err:.2e}")
    # nearest-2x folding pre-sums 2-4 taps and rounds the sums to fp16: a weight rounding (2^-12 rms) the 9-tap form does not have
    # fp32 accumulation order over K up to 11,520 terms: 1e-5.  With the GroupNorm prologue the kernel's SiLU (v_exp / v_rcp, 1 ulp)
    # and torch's differ in the last fp32 bit, which now and then flips the fp16 rounding of an operand element: 1e-4
    tol = 6e-4 if ups else (1e-4 if use_gn else 1e-5)
    assert torch.isfinite(got).all() and err <= tol, f"{name}: {err:.3e}"
    if R > 0:   # fused statistics describe the fp32 value (hi + lo), not the rounded hi half
        gamma, beta = torch.ones(Cout), torch.zeros(Cout)
        scale, shift = torch.empty((B, Cout), device=DEV), torch.empty((B, Cout), device=DEV)
        gd, btd = gamma.to(DEV), beta.to(DEV)
        _lib.check(lib.ldiff_op_gn_finalize(st.data_ptr(), R, Cout, None, 0, 0, B, Ho * Wo, 32, 1e-5, gd.data_ptr(), btd.data_ptr(),
                                            scale.data_ptr(), shift.data_ptr(), sp()))
        torch.cuda.synchronize()
        gotn = got.float() * scale.cpu()[:, :, None, None] + shift.cpu()[:, :, None, None]
        refn = F.group_norm(got.float(), 32, gamma, beta, 1e-5)
        assert (gotn - refn).abs().max() <= 2e-4 * max(1.0, refn.abs().max())


@pytest.mark.parametrize("shape", ["one_unit_one_slab", "two_units_one_slab", "three_units_two_slabs", "two_channel_tiles_four_slabs",
                                   "four_channel_tiles_eight_slabs", "four_channel_tiles_eight_slabs_short_runs", "three_units_two_slabs_short_runs"])
@pytest.mark.parametrize("res,stats,temb", [(0, 0, 0), (1, 0, 1), (0, 1, 0), (1, 1, 1)])
def test_conv3x3_dataflow_kernel(lib, shape, res, stats, temb):
    """The producer / consumer conv3x3 kernel (kernels_conv3x3d.hip: GroupNorm + SiLU prologue on maps that fill the chip): every epilogue
    configuration (residual x fused statistics, with and without a time embedding) on unit lists with one unit per workgroup, several
    units (tile switches, bias table hand-over) and several channel tiles / slabs.  Three launches each: producers and consumers meet only
    through progress words in LDS, and a missing wait would show up as a now-and-then wrong tile.  The statistics are checked against the
    sums of the kernel's own fp16 outputs (they are defined on the rounded values)."""
    # four_channel_tiles_eight_slabs = the VAE decoder's ten 512 -> 512 convs on the 64 x 64 maps at B = 8 (the bench configuration: 512 units >= 256 CUs
    # only at that batch); *_short_runs = one unit per workgroup, as ldiff_sample launches the decodes beside the next UNet pass
    short = shape.endswith("_short_runs")
    B, Cin, H, W, Cout = {"one_unit_one_slab": (1, 64, 256, 256, 128), "two_units_one_slab": (3, 64, 256, 256, 128), "three_units_two_slabs": (3, 128, 256, 256, 128),
                          "two_channel_tiles_four_slabs": (2, 256, 128, 128, 256), "four_channel_tiles_eight_slabs": (8, 512, 64, 64, 512)}[shape.replace("_short_runs", "")]
    g = torch.Generator().manual_seed(res * 4 + stats * 2 + temb + len(shape))
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.float16)
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16)
    bias = torch.randn(Cout, generator=g) * 0.1
    sc, sh = 1.0 + 0.2 * torch.randn((B, Cin), generator=g), 0.2 * torch.randn((B, Cin), generator=g)
    a = F.silu(x.float() * sc[:, None, None, :] + sh[:, None, None, :]).to(torch.float16).float()   # the kernel rounds the normalised operand to fp16
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1)
    xd, wd, bd, scd, shd = x.to(DEV), w.reshape(Cout, -1).contiguous().to(DEV), bias.to(DEV), sc.to(DEV), sh.to(DEV)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = xd.data_ptr(), Cin, B, H, W, H, W
    a_.ks, a_.stride, a_.pad_t, a_.pad_l = 3, 1, 1, 1
    a_.w, a_.N, a_.Nrows, a_.bias = wd.data_ptr(), Cout, Cout, bd.data_ptr()
    a_.gn_scale, a_.gn_shift, a_.silu_in = scd.data_ptr(), shd.data_ptr(), 1
    a_.short_runs = 1 if short else 0
    y = torch.empty((B, H, W, Cout), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy = y.data_ptr(), Cout
    if temb:
        t = torch.randn((B, Cout), generator=g) * 0.3
        td = t.to(DEV)
        a_.temb, a_.ld_temb = td.data_ptr(), Cout
        ref = ref + t[:, :, None, None]
    if res:
        r = (torch.randn((B, H, W, Cout), generator=g) * 2.0).to(torch.float16)
        rd = r.to(DEV)
        a_.res, a_.ld_res = rd.data_ptr(), Cout
        ref = ref + r.float().permute(0, 3, 1, 2)
    if stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
        assert R > 0
        st = torch.empty((B, Cout, R, 2), device=DEV)
        a_.stats = st.data_ptr()
    for it in range(3):
        y.fill_(float("nan"))
        if stats:
            st.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        torch.cuda.synchronize()
        got = y.float().cpu().permute(0, 3, 1, 2)
        assert_close(got, ref, f"{shape} launch {it}")
        if stats:
            sums = st.double().cpu().sum(dim=2)                       # [B, Cout, 2]
            yd = y.double().cpu()
            want = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=-1)
            assert torch.isfinite(sums).all(), f"launch {it}: statistics not written everywhere"
            err = ((sums - want).abs() / (want.abs() + H * W * 1e-3)).max().item()
            assert err <= 1e-4, f"launch {it}: fused statistics differ from the sums of the outputs by {err:.3e}"


@pytest.mark.parametrize("shape", ["two_slabs_one_channel_tile", "four_slabs_two_channel_tiles", "two_slabs_one_channel_tile_short_runs", "one_slab"])
@pytest.mark.parametrize("stats", [0, 1])
def test_conv3x3_dataflow_kernel_upsample(lib, shape, stats):
    """Upsample2D (nearest 2x) + conv3x3 of the VAE decoder on the producer / consumer kernel (ups = 1: four parity units per pixel tile, four pre-summed
    taps each, raw operand, outputs interleaved into the 2H x 2W map) against torch's interpolate + conv2d; the fused statistics against the sums of the
    kernel's own outputs; three launches each, bit-identical (progress words: a missing wait shows up as a now-and-then wrong tile)."""
    short = shape.endswith("_short_runs")
    B, Cin, H, W, Cout = {"two_slabs_one_channel_tile": (2, 128, 128, 128, 128), "four_slabs_two_channel_tiles": (1, 256, 64, 128, 256),
                          "one_slab": (2, 64, 128, 128, 128)}[shape.replace("_short_runs", "")]
    g = torch.Generator().manual_seed(stats + len(shape))
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.float16)
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16)
    bias = torch.randn(Cout, generator=g) * 0.1
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(up, w.float().permute(0, 3, 1, 2), bias, padding=1)
    xd, wd, bd = x.to(DEV), w.reshape(Cout, -1).contiguous().to(DEV), bias.to(DEV)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = xd.data_ptr(), Cin, B, H, W, 2 * H, 2 * W
    a_.ks, a_.stride, a_.pad_t, a_.pad_l, a_.ups, a_.c3d_ups = 3, 1, 1, 1, 1, 1
    a_.w, a_.N, a_.Nrows, a_.bias = wd.data_ptr(), Cout, Cout, bd.data_ptr()
    a_.short_runs = 1 if short else 0
    y = torch.empty((B, 2 * H, 2 * W, Cout), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy = y.data_ptr(), Cout
    if stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
        assert R > 0
        st = torch.empty((B, Cout, R, 2), device=DEV)
        a_.stats = st.data_ptr()
    first = None
    for it in range(3):
        y.fill_(float("nan"))
        if stats:
            st.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        torch.cuda.synchronize()
        # (the four taps of a parity are sums of up to four fp16 weights, rounded to fp16 once more: 6e-4, as for the other parity-folded kernels)
        assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, f"{shape} launch {it}")
        if first is None:
            first = y.clone()
        assert torch.equal(y, first), f"launch {it} differs from the first"
        if stats:
            sums = st.double().cpu().sum(dim=2)
            yd = y.double().cpu()
            want = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=-1)
            assert torch.isfinite(sums).all(), f"launch {it}: statistics not written everywhere"
            assert ((sums - want).abs() / (want.abs() + 4 * H * W * 1e-3)).max().item() <= 1e-4


@pytest.mark.parametrize("shape", ["vae_256_to_128", "vae_512_to_256", "one_slab_each", "vae_256_to_128_short_runs", "pitched_source"])
@pytest.mark.parametrize("stats", [0, 1])
def test_conv3x3_dataflow_kernel_with_folded_shortcut(lib, shape, stats):
    """ldiff_conv_args.sc_*: the 1x1 conv_shortcut of a width-changing ResnetBlock2D as extra centre-tap slabs of the block's second conv on the
    producer / consumer kernel (raw operand, no GroupNorm; weights behind the nine taps; biases summed): y = conv3x3(silu(gn(x))) + sc_w . sc_x + b + b_sc,
    against torch fp32 on the operands the kernel sees.  Persistent runs of several units and one unit per workgroup, with and without the fused
    statistics, a source read through a row pitch; three launches each (progress words: a missing wait shows up as a now-and-then wrong tile)."""
    short = shape.endswith("_short_runs")
    B, Cin, H, W, Cout, Cs = {"vae_256_to_128": (3, 128, 256, 256, 128, 256), "vae_512_to_256": (2, 256, 128, 128, 256, 512), "one_slab_each": (2, 64, 256, 256, 128, 64),
                              "pitched_source": (2, 128, 256, 256, 128, 192)}[shape.replace("_short_runs", "")]
    g = torch.Generator().manual_seed(stats + len(shape))
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.float16)
    pitch = Cs + 64 if shape == "pitched_source" else Cs
    xs_full = torch.randn((B, H, W, pitch), generator=g).to(torch.float16)
    xs = xs_full[..., :Cs]
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16)
    wsc = (torch.randn((Cout, Cs), generator=g) / math.sqrt(Cs)).to(torch.float16)
    bias, bsc = torch.randn(Cout, generator=g) * 0.1, torch.randn(Cout, generator=g) * 0.1
    sc, sh = 1.0 + 0.2 * torch.randn((B, Cin), generator=g), 0.2 * torch.randn((B, Cin), generator=g)
    a = F.silu(x.float() * sc[:, None, None, :] + sh[:, None, None, :]).to(torch.float16).float()
    ref = F.conv2d(a.permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1)
    ref = ref + F.conv2d(xs.float().permute(0, 3, 1, 2), wsc.float()[:, :, None, None], bsc)
    xd, xsd, wd, wscd = x.to(DEV), xs_full.to(DEV), w.reshape(Cout, -1).contiguous().to(DEV), wsc.contiguous().to(DEV)
    bd, bscd, scd, shd = bias.to(DEV), bsc.to(DEV), sc.to(DEV), sh.to(DEV)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = xd.data_ptr(), Cin, B, H, W, H, W
    a_.ks, a_.stride, a_.pad_t, a_.pad_l = 3, 1, 1, 1
    a_.w, a_.N, a_.Nrows, a_.bias = wd.data_ptr(), Cout, Cout, bd.data_ptr()
    a_.gn_scale, a_.gn_shift, a_.silu_in = scd.data_ptr(), shd.data_ptr(), 1
    a_.short_runs = 1 if short else 0
    a_.sc_x, a_.sc_C, a_.sc_ld, a_.sc_w, a_.sc_bias = xsd.data_ptr(), Cs, pitch, wscd.data_ptr(), bscd.data_ptr()
    y = torch.empty((B, H, W, Cout), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy = y.data_ptr(), Cout
    if stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
        assert R > 0
        st = torch.empty((B, Cout, R, 2), device=DEV)
        a_.stats = st.data_ptr()
    first = None
    for it in range(3):
        y.fill_(float("nan"))
        if stats:
            st.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        torch.cuda.synchronize()
        assert_close(y.float().cpu().permute(0, 3, 1, 2), ref, f"{shape} launch {it}")
        if first is None:
            first = y.clone()
        assert torch.equal(y, first), f"launch {it} differs from the first"
        if stats:
            sums = st.double().cpu().sum(dim=2)
            yd = y.double().cpu()
            want = torch.stack([yd.sum(dim=(1, 2)), (yd * yd).sum(dim=(1, 2))], dim=-1)
            assert torch.isfinite(sums).all()
            assert ((sums - want).abs() / (want.abs() + H * W * 1e-3)).max().item() <= 1e-4
    # what it replaces: the shortcut as its own 1x1 launch, its fp16 output as the residual of the 3x3 conv (two roundings more): same result to fp16 noise
    ysc = torch.empty((B, H, W, Cout), dtype=torch.float16, device=DEV)
    b_ = _lib.ConvArgs()
    b_.x, b_.C1, b_.ld1, b_.B, b_.Hin, b_.Win, b_.Hout, b_.Wout, b_.ks, b_.stride = xsd.data_ptr(), Cs, pitch, B, H, W, H, W, 1, 1
    b_.w, b_.N, b_.Nrows, b_.bias, b_.y, b_.ldy = wscd.data_ptr(), Cout, Cout, bscd.data_ptr(), ysc.data_ptr(), Cout
    _lib.check(lib.ldiff_op_conv(C.byref(b_), sp()))
    a_.sc_x, a_.sc_w, a_.sc_bias, a_.sc_C, a_.sc_ld = None, None, None, 0, 0
    a_.res, a_.ld_res, a_.stats = ysc.data_ptr(), Cout, None
    y2 = torch.empty_like(y)
    a_.y = y2.data_ptr()
    _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
    torch.cuda.synchronize()
    assert (y2.float() - first.float()).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    # shapes the kernel does not take are refused, not silently run without the shortcut
    a_.res, a_.ld_res, a_.sc_x, a_.sc_C, a_.sc_ld, a_.sc_w = None, 0, xsd.data_ptr(), Cs - 32, pitch, wscd.data_ptr()
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))


@pytest.mark.parametrize("shape", ["one_tile_per_workgroup", "two_tiles_per_workgroup"])
@pytest.mark.parametrize("res,lo,stats", [(r, l, s_) for r in (0, 1) for l in (0, 1) for s_ in (0, 1)])
def test_conv3x3_persistent_kernel_epilogue_configs(lib, shape, res, lo, stats):
    """Every epilogue configuration the persistent 16x16-tile kernel is instantiated for (residual x split output x fused statistics), on
    a tile list with exactly one tile per workgroup (prologue -> last step -> epilogue with nothing in between) and with two.  Three
    launches each: the kernel's two wave groups run half a step apart, and a missing wait shows up as a now-and-then wrong patch."""
    B, Cin, H, W, Cout = (1, 64, 256, 256, 128) if shape == "one_tile_per_workgroup" else (8, 64, 128, 128, 128)
    g = torch.Generator().manual_seed(res * 4 + lo * 2 + stats)
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.float16)
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16)
    bias = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1)
    xd, wd, bd = x.to(DEV), w.reshape(Cout, -1).contiguous().to(DEV), bias.to(DEV)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = xd.data_ptr(), Cin, B, H, W, H, W
    a_.ks, a_.stride, a_.pad_t, a_.pad_l = 3, 1, 1, 1
    a_.w, a_.N, a_.Nrows, a_.bias = wd.data_ptr(), Cout, Cout, bd.data_ptr()
    ld = 2 * Cout if lo else Cout
    y = torch.empty((B, H, W, ld), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy, a_.y_lo = y.data_ptr(), ld, Cout if lo else 0
    if res:
        r = (torch.randn((B, H, W, Cout), generator=g) * 2.0).to(torch.float16)
        rd = r.to(DEV)
        a_.res, a_.ld_res = rd.data_ptr(), Cout
        ref = ref + r.float().permute(0, 3, 1, 2)
    if stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
        assert R > 0
        st = torch.empty((B, Cout, R, 2), device=DEV)
        a_.stats = st.data_ptr()
    gamma, beta = torch.ones(Cout), torch.zeros(Cout)
    for it in range(3):
        y.fill_(float("nan"))
        if stats:
            st.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        torch.cuda.synchronize()
        yc = y.float().cpu()
        got = (yc[..., :Cout] + (yc[..., Cout:] if lo else 0.0)).permute(0, 3, 1, 2)
        assert torch.isfinite(got).all(), f"launch {it}: non-finite output"
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err <= (2e-5 if lo else 1e-3), f"launch {it}: rel err {err:.3e}"
        if lo:   # the lo half is a rounding remainder: anything larger is a sum that was read before it was complete
            assert yc[..., Cout:].abs().max() <= 2.0 ** -10 * max(1.0, yc[..., :Cout].abs().max().item())
        if stats:
            scale, shift = torch.empty((B, Cout), device=DEV), torch.empty((B, Cout), device=DEV)
            gd, btd = gamma.to(DEV), beta.to(DEV)
            _lib.check(lib.ldiff_op_gn_finalize(st.data_ptr(), R, Cout, None, 0, 0, B, H * W, 32, 1e-5, gd.data_ptr(), btd.data_ptr(),
                                                scale.data_ptr(), shift.data_ptr(), sp()))
            torch.cuda.synchronize()
            seen = got if lo else yc[..., :Cout].permute(0, 3, 1, 2)   # statistics describe what the consumer reads
            gotn = seen * scale.cpu()[:, :, None, None] + shift.cpu()[:, :, None, None]
            refn = F.group_norm(seen, 32, gamma, beta, 1e-5)
            assert (gotn - refn).abs().max() <= 2e-4 * max(1.0, refn.abs().max().item()), f"launch {it}: fused statistics"


@pytest.mark.parametrize("Cin,res", [(128, 0), (128, 1), (256, 1)])
def test_conv3x3_split_operand_with_fp8_lo_half(lib, Cin, res):
    """Split conv operand whose lo half is fp8 (ldiff_op_norm_apply_lo8 -> [C fp16 | C e4m3 of lo * 2^15], ldiff_op_lo8_weights ->
    [C fp16 | C e4m3 of w * 2^sw] per tap, ldiff_conv_args.lo8_slab0): the 16 x 16 ping-pong kernel runs the C / 64 fp16 slabs through the fp16
    MFMA and the C / 128 fp8 slabs through the block-scaled fp8 MFMA.  Checked (a) against the exact statement of what the kernel is given --
    conv(hi, w) + conv(decoded lo8, decoded w8) in fp64, to fp32 round-off -- and (b) against the conv over the fp32 operand: ~2^-14 instead of
    the 2^-11 of a plain fp16 operand.  Split output with fused statistics, with and without a split residual; three launches (ping-pong)."""
    B, H, W, Cout = 4, 128, 128, 128
    g = torch.Generator().manual_seed(Cin + res)
    x = torch.randn((B, H, W, Cin), generator=g) * 1.5
    x[0, 3, 5, 7] = 90.0                                                   # its lo (2^-5) saturates e4m3 at 448 / 2^15: degrades gracefully
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16)
    bias = torch.randn(Cout, generator=g) * 0.1
    # producer: GroupNorm-apply with scale 1, shift 0, no activation = the plain split of x
    x32 = torch.cat([x.to(torch.float16), (x - x.to(torch.float16).float()).to(torch.float16)], -1).contiguous().to(DEV)   # a split source (hi | lo)
    ones, zeros = torch.ones((B, Cin), device=DEV), torch.zeros((B, Cin), device=DEV)
    xq = torch.empty((B, H, W, 3 * Cin), dtype=torch.uint8, device=DEV)
    _lib.check(lib.ldiff_op_norm_apply_lo8(x32.data_ptr(), Cin, 2 * Cin, Cin, B, H * W, ones.data_ptr(), zeros.data_ptr(), 0, xq.data_ptr(), sp()))
    wd = w.reshape(Cout, -1).contiguous().to(DEV)
    wq = torch.empty((Cout, 9, 3 * Cin), dtype=torch.uint8, device=DEV)
    wsc = torch.zeros(4, dtype=torch.int32, device=DEV)
    _lib.check(lib.ldiff_op_lo8_weights(wd.data_ptr(), wq.data_ptr(), wsc.data_ptr(), Cout, 9, Cin, sp()))
    torch.cuda.synchronize()
    # decode what the kernel is given
    xqc, wqc, sw = xq.cpu(), wq.cpu(), 127 - int(wsc[0].item())
    x_hi = xqc[..., :2 * Cin].contiguous().view(torch.float16).double()
    x_lo = xqc[..., 2 * Cin:].contiguous().view(torch.float8_e4m3fn).float().double() * 2.0 ** -15
    w_hi = wqc[..., :2 * Cin].contiguous().view(torch.float16).double().reshape(Cout, 3, 3, Cin)
    w_lo = wqc[..., 2 * Cin:].contiguous().view(torch.float8_e4m3fn).float().double().reshape(Cout, 3, 3, Cin) * 2.0 ** -sw
    assert torch.equal(w_hi.float(), w.float().reshape(Cout, 3, 3, Cin)), "hi halves of the weights are the fp16 weights"
    assert (x_hi.float() != x.to(torch.float16).float()).float().mean() <= 1e-3   # (hi + lo re-rounded by the producer: equal up to ties)
    assert 2.0 ** sw * w.float().abs().max() <= 448.0 < 2.0 ** (sw + 1) * w.float().abs().max()
    lo_true = x.double() - x_hi
    assert ((x_lo - lo_true).abs() <= 2.0 ** -4 * lo_true.abs() + 2.0 ** -25).sum() >= x_lo.numel() - 1   # three mantissa bits (one element saturated)
    nchw = lambda t: t.permute(0, 3, 1, 2)
    ref_given = F.conv2d(nchw(x_hi), nchw(w_hi), bias.double(), padding=1) + F.conv2d(nchw(x_lo), nchw(w_lo), None, padding=1)
    ref_full = F.conv2d(nchw(x.double()), nchw(w.double()), bias.double(), padding=1)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout = xq.data_ptr(), Cin + Cin // 2, B, H, W, H, W
    a_.ks, a_.stride, a_.pad_t, a_.pad_l = 3, 1, 1, 1
    a_.w, a_.N, a_.Nrows, a_.bias = wq.data_ptr(), Cout, Cout, bias.to(DEV).data_ptr()
    bias_d = bias.to(DEV); a_.bias = bias_d.data_ptr()
    a_.lo8_slab0, a_.lo8_scale = Cin // 64, wsc.data_ptr()
    y = torch.empty((B, H, W, 2 * Cout), dtype=torch.float16, device=DEV)
    a_.y, a_.ldy, a_.y_lo = y.data_ptr(), 2 * Cout, Cout
    if res:
        r = torch.randn((B, H, W, Cout), generator=g) * 2.0
        rd = torch.cat([r.to(torch.float16), (r - r.to(torch.float16).float()).to(torch.float16)], -1).contiguous().to(DEV)
        a_.res, a_.ld_res, a_.res_lo = rd.data_ptr(), 2 * Cout, Cout
        radd = nchw(rd.cpu()[..., :Cout].double() + rd.cpu()[..., Cout:].double())
        ref_given, ref_full = ref_given + radd, ref_full + radd
    R = lib.ldiff_op_conv_stats_blocks(C.byref(a_))
    assert R > 0
    st = torch.empty((B, Cout, R, 2), device=DEV)
    a_.stats = st.data_ptr()
    for it in range(3):
        y.fill_(float("nan")); st.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a_), sp()))
        torch.cuda.synchronize()
        yc = y.cpu().double()
        got = nchw(yc[..., :Cout] + yc[..., Cout:])
        assert torch.isfinite(got).all(), f"launch {it}: non-finite output"
        e_given = ((got - ref_given).abs().max() / ref_given.abs().max()).item()
        e_full = ((got - ref_full).abs().max() / ref_full.abs().max()).item()
        print(f"Cin {Cin} res {res} launch {it}: against the given operands {e_given:.2e}, against the fp32 operand {e_full:.2e} of range")
        assert e_given <= 3e-6, f"launch {it}: {e_given:.3e} from the exact statement of the given operands"
        assert e_full <= 4e-5, f"launch {it}: {e_full:.3e} from the conv over the fp32 operand"
        sums = st.double().cpu().sum(dim=2)
        want = torch.stack([got.sum(dim=(2, 3)), (got * got).sum(dim=(2, 3))], dim=-1)
        assert torch.isfinite(sums).all() and ((sums - want).abs() / (want.abs() + H * W * 1e-3)).max() <= 1e-4, f"launch {it}: fused statistics"


def test_cu_share_stream_runs_the_same_kernels(lib):
    """ldiff_stream_create_cu_share: a stream restricted to a quarter of every XCD's CUs runs a conv and an attention launch to the same bits as
    the whole-chip stream (the measurement aid behind profiles/r04_cu_partition.txt); bad shares are rejected."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn((2, 64, 32, 32), generator=g)
    w = torch.randn((64, 64, 3, 3), generator=g) / 24.0
    xd = nhwc16(x)
    wd = w.permute(0, 2, 3, 1).to(torch.float16).reshape(64, -1).contiguous().to(DEV)
    a_ = _lib.ConvArgs()
    a_.x, a_.C1, a_.B, a_.Hin, a_.Win, a_.Hout, a_.Wout, a_.ks, a_.stride, a_.pad_t, a_.pad_l = xd.data_ptr(), 64, 2, 32, 32, 32, 32, 3, 1, 1, 1
    a_.w, a_.N, a_.Nrows = wd.data_ptr(), 64, 64
    outs = []
    raw = C.c_void_p()
    _lib.check(lib.ldiff_stream_create_cu_share(8, 16, C.byref(raw)))
    for stream in (sp(), raw):
        y = torch.full((2, 32, 32, 64), float("nan"), dtype=torch.float16, device=DEV)
        a_.y, a_.ldy = y.data_ptr(), 64
        _lib.check(lib.ldiff_op_conv(C.byref(a_), stream))
        torch.cuda.synchronize()
        outs.append(y.clone())
    _lib.check(lib.ldiff_stream_destroy(raw))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    bad = C.c_void_p()
    assert lib.ldiff_stream_create_cu_share(4, 16, C.byref(bad)) != 0      # not a multiple of 8
    assert lib.ldiff_stream_create_cu_share(16, 16, C.byref(bad)) != 0     # empty share


def test_split_operand_beats_plain_operand(lib):
    """The point of the split operand: the same 1x1 conv over the same fp32 stream is ~2^-11 accurate with a plain fp16 operand and
    ~1e-6 with the split one."""
    g = torch.Generator().manual_seed(1)
    M, Cc, N = 512, 320, 320
    x = torch.randn((1, 1, M, Cc), generator=g)
    w = r16(torch.randn((N, Cc), generator=g) / math.sqrt(Cc))
    ref = (x[0, 0].double() @ w.double().t())
    wd = w.to(torch.float16).contiguous().to(DEV)
    wdup = torch.empty((N, 2 * Cc), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_dup_weights(wd.data_ptr(), wdup.data_ptr(), N, 1, Cc, Cc, 0, 2 * Cc, sp()))
    xs = to_split(x).to(DEV)
    errs = {}
    for mode in ("plain", "split"):
        a = _lib.ConvArgs()
        a.x, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride = xs.data_ptr(), 1, 1, M, 1, M, 1, 1
        if mode == "plain":
            a.C1, a.ld1, a.w = Cc, 2 * Cc, wd.data_ptr()
        else:
            a.C1, a.w = 2 * Cc, wdup.data_ptr()
        y = torch.empty((M, N), dtype=torch.float32, device=DEV)
        a.N, a.Nrows, a.y, a.ldy, a.out_f32 = N, N, y.data_ptr(), N, 1
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp()))
        torch.cuda.synchronize()
        errs[mode] = ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item()
    print(f"1x1 conv over an fp32 stream: plain fp16 operand {errs['plain']:.2e}, split operand {errs['split']:.2e}")
    assert errs["split"] <= 3e-6 and errs["plain"] > 20 * errs["split"]


@pytest.mark.parametrize("C1,C2,in_split,out_split,silu", [(64, 0, True, True, 1), (128, 64, True, False, 1), (320, 0, False, True, 0), (64, 32, True, True, 0)])
def test_norm_apply(lib, C1, C2, in_split, out_split, silu):
    g = torch.Generator().manual_seed(C1 + C2)
    B, HW = 2, 300
    Cc = C1 + C2
    x = torch.randn((B, HW, Cc), generator=g) * 2
    scale, shift = 1 + 0.2 * torch.randn((B, Cc), generator=g), 0.3 * torch.randn((B, Cc), generator=g)
    pack = to_split if in_split else (lambda t: t.to(torch.float16).contiguous())
    x1d = pack(x[..., :C1]).to(DEV)
    x2d = pack(x[..., C1:]).to(DEV) if C2 else None
    seen = (lambda t, C: from_split(t.cpu(), C)) if in_split else (lambda t, C: t.cpu().float())
    xin = seen(x1d, C1) if not C2 else torch.cat([seen(x1d, C1), seen(x2d, C2)], -1)
    ref = xin.double() * scale.double()[:, None, :] + shift.double()[:, None, :]
    if silu:
        ref = F.silu(ref)
    ld = lambda C: 2 * C if in_split else C
    lo = lambda C: C if in_split else 0
    y = torch.full((B, HW, 2 * Cc if out_split else Cc), float("nan"), dtype=torch.float16, device=DEV)
    sd, hd = scale.to(DEV), shift.to(DEV)
    _lib.check(lib.ldiff_op_norm_apply(x1d.data_ptr(), C1, ld(C1), lo(C1), x2d.data_ptr() if C2 else None, C2, ld(C2), lo(C2), B, HW, sd.data_ptr(),
                                       hd.data_ptr(), silu, y.data_ptr(), y.shape[-1], Cc if out_split else 0, sp()))
    torch.cuda.synchronize()
    yc = y.cpu()
    if out_split:
        err = (from_split(yc, Cc).double() - ref).abs().max() / ref.abs().max()
        assert err <= 3e-6, f"split output err {err:.2e}"
    else:
        assert_close(yc, ref.float(), "norm_apply", rtol=6e-4, atol_rel=1e-6)   # one fp16 rounding of the exact value
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_op_norm_apply(x1d.data_ptr(), C1, ld(C1), lo(C1), None, 0, 0, 0, B, HW, sd.data_ptr(), hd.data_ptr(), silu, y.data_ptr(), C1 - 8, 0, sp()))


@pytest.mark.parametrize("rows,Cc", [(1027, 640), (515, 1280), (9, 320), (8192, 640)])
def test_layernorm_on_split_rows_at_the_unet_widths(lib, rows, Cc):
    """LayerNorm of split rows hi | lo at C = 320 / 640 / 1280, ragged row counts, against fp32 LayerNorm of hi + lo with ONE fp16 rounding; nothing
    is written behind the last row."""
    g = torch.Generator().manual_seed(rows + Cc)
    x = torch.randn((rows, Cc), generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    xs = to_split(x).to(DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    y = torch.full((rows + 3, Cc), float("nan"), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_layernorm(xs.data_ptr(), 2 * Cc, Cc, y.data_ptr(), rows, Cc, gd.data_ptr(), bd.data_ptr(), 1e-5, sp()))
    torch.cuda.synchronize()
    ref = F.layer_norm(from_split(xs.cpu(), Cc), (Cc,), gamma, beta, 1e-5)
    assert_close(y[:rows], ref, f"layernorm split {rows}x{Cc}", rtol=6e-4, atol_rel=1e-5)
    assert torch.isnan(y[rows:].float()).all(), "rows beyond the tensor were written"


def test_layernorm_and_groupnorm_statistics_read_split_tensors(lib):
    g = torch.Generator().manual_seed(12)
    rows, Cc = 513, 320
    x = torch.randn((rows, Cc), generator=g) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    xs = to_split(x).to(DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    y = torch.empty((rows, Cc), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_layernorm(xs.data_ptr(), 2 * Cc, Cc, y.data_ptr(), rows, Cc, gd.data_ptr(), bd.data_ptr(), 1e-5, sp()))
    torch.cuda.synchronize()
    ref = F.layer_norm(from_split(xs.cpu(), Cc), (Cc,), gamma, beta, 1e-5)
    assert_close(y, ref, "layernorm split", rtol=6e-4, atol_rel=1e-5)           # exactly one fp16 rounding
    # hi-only view of the same buffer (pitch 2C, lo = 0) == LayerNorm of the rounded tensor
    _lib.check(lib.ldiff_op_layernorm(xs.data_ptr(), 2 * Cc, 0, y.data_ptr(), rows, Cc, gd.data_ptr(), bd.data_ptr(), 1e-5, sp()))
    torch.cuda.synchronize()
    assert_close(y, F.layer_norm(r16(x), (Cc,), gamma, beta, 1e-5), "layernorm pitch", rtol=1e-3, atol_rel=1e-3)
    # GroupNorm statistics over a split concat
    B, HW, C1, C2 = 2, 256, 128, 64
    xg = torch.randn((B, HW, C1 + C2), generator=g) * 1.7 + 0.3
    x1, x2 = to_split(xg[..., :C1]).to(DEV), to_split(xg[..., C1:]).to(DEV)
    gam, bet = 1 + 0.1 * torch.randn(C1 + C2, generator=g), 0.1 * torch.randn(C1 + C2, generator=g)
    scale, shift = torch.empty((B, C1 + C2), device=DEV), torch.empty((B, C1 + C2), device=DEV)
    g2, b2 = gam.to(DEV), bet.to(DEV)
    _lib.check(lib.ldiff_op_gn_stats(x1.data_ptr(), C1, 2 * C1, C1, x2.data_ptr(), C2, 2 * C2, C2, B, HW, 32, 1e-5, g2.data_ptr(), b2.data_ptr(),
                                     scale.data_ptr(), shift.data_ptr(), sp()))
    torch.cuda.synchronize()
    xr = torch.cat([from_split(x1.cpu(), C1), from_split(x2.cpu(), C2)], -1)
    got = xr * scale.cpu()[:, None, :] + shift.cpu()[:, None, :]
    ref = F.group_norm(xr.permute(0, 2, 1), 32, gam, bet, 1e-5).permute(0, 2, 1)
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, ref.abs().max())



# ======================================================================================================================
# LayerNorm folded into the consuming linear layer (kernels_gemm_ast.hip): activations stationary in registers
# ======================================================================================================================
def _geglu_perm(inner):
    perm = torch.empty(2 * inner, dtype=torch.long)
    for r in range(2 * inner):
        q = r if r < inner else r - inner
        perm[(q // 16) * 32 + (0 if r < inner else 16) + q % 16] = r
    return perm


@pytest.mark.parametrize("M,N,split,geglu", [(512, 960, True, False), (300, 320, True, False), (1024, 2560, True, True), (256, 64, False, False),
                                             (777, 320, False, False), (33, 256, False, True), (8192, 960, True, False)])
def test_ln_linear_fused(lib, M, N, split, geglu):
    """y = LayerNorm(x) W^T + bias (+ GEGLU) in ONE launch (ldiff_op_ln_linear) against torch: LayerNorm of the fp32 value (hi + lo of a split
    tensor), ONE fp16 rounding of the normalised operand, fp32 accumulation -- the arithmetic of the two-launch form (ldiff_op_layernorm +
    ldiff_op_conv) it replaces on the C = 320 level of the UNet (norm1 -> to_q/k/v, norm2 -> attn2.to_q, norm3 -> ff.net.0.proj).  Row counts
    that do not fill the 256-row workgroup, odd panel counts (column splits of unequal length), ragged last rows."""
    Cc = 320
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn((M, Cc), generator=g) * 2 + 0.5
    x[0] *= 30.0                                                                    # a row far from the others' scale
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    w = torch.randn((N, Cc), generator=g) / math.sqrt(Cc)
    b = torch.randn(N, generator=g) * 0.2
    xs = to_split(x) if split else x.to(torch.float16)
    xv = from_split(xs, Cc) if split else xs.float()
    a16 = r16(F.layer_norm(xv, (Cc,), gamma, beta, 1e-5))
    proj = a16 @ r16(w).t() + b
    if geglu:
        inner = N // 2
        ref = proj[:, :inner] * F.gelu(proj[:, inner:])
        perm = _geglu_perm(inner)
        w, b = w[perm], b[perm]
    else:
        ref = proj
    xd, wd, bd, gd, btd = xs.to(DEV), w.to(torch.float16).contiguous().to(DEV), b.contiguous().to(DEV), gamma.to(DEV), beta.to(DEV)
    Nout = N // 2 if geglu else N
    for bias in (bd, None):
        y = torch.full((M, Nout), float("nan"), dtype=torch.float16, device=DEV)
        _lib.check(lib.ldiff_op_ln_linear(xd.data_ptr(), 2 * Cc if split else Cc, Cc if split else 0, M, Cc, gd.data_ptr(), btd.data_ptr(), 1e-5, wd.data_ptr(), N, N,
                                          None if bias is None else bias.data_ptr(), 1 if geglu else 0, y.data_ptr(), Nout, 0, 1.0, sp()))
        torch.cuda.synchronize()
        if bias is None:
            pr = a16 @ r16(w).t()
            want = pr[:, :N // 2] * F.gelu(pr[:, N // 2:]) if False else None
            if geglu:
                p0 = a16 @ r16(w[torch.argsort(_geglu_perm(N // 2))]).t()
                want = p0[:, :N // 2] * F.gelu(p0[:, N // 2:])
            else:
                want = pr
        else:
            want = ref
        # the normalised operand may round differently in the last fp16 bit where the kernel's fp32 LayerNorm and torch's differ by an ulp
        assert_close(y, want, f"ln_linear M={M} N={N} split={split} geglu={geglu} bias={bias is not None}", rtol=2e-3, atol_rel=1.5e-3)
    # equal to the two-launch form to fp32 accumulation order
    n = torch.empty((M, Cc), dtype=torch.float16, device=DEV)
    _lib.check(lib.ldiff_op_layernorm(xd.data_ptr(), 2 * Cc if split else Cc, Cc if split else 0, n.data_ptr(), M, Cc, gd.data_ptr(), btd.data_ptr(), 1e-5, sp()))
    y2 = torch.full((M, Nout), float("nan"), dtype=torch.float16, device=DEV)
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride = n.data_ptr(), Cc, 1, 1, M, 1, M, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy, a.geglu = wd.data_ptr(), N, N, bd.data_ptr(), y2.data_ptr(), Nout, 1 if geglu else 0
    _lib.check(lib.ldiff_op_conv(C.byref(a), sp()))
    y1 = torch.empty_like(y2)
    _lib.check(lib.ldiff_op_ln_linear(xd.data_ptr(), 2 * Cc if split else Cc, Cc if split else 0, M, Cc, gd.data_ptr(), btd.data_ptr(), 1e-5, wd.data_ptr(), N, N,
                                      bd.data_ptr(), 1 if geglu else 0, y1.data_ptr(), Nout, 0, 1.0, sp()))
    torch.cuda.synchronize()
    d = (y1.float() - y2.float()).abs().max().item()
    assert d <= 2e-3 * max(1.0, y2.float().abs().max().item()), f"fused vs two launches: {d:.3e}"
    with pytest.raises(ValueError):     # shapes the kernel does not take are refused, not silently mis-computed
        _lib.check(lib.ldiff_op_ln_linear(xd.data_ptr(), 2 * Cc if split else Cc, Cc if split else 0, M, Cc, gd.data_ptr(), btd.data_ptr(), 1e-5, wd.data_ptr(), 48, N,
                                          bd.data_ptr(), 0, y1.data_ptr(), Nout, 0, 1.0, sp()))


def test_ln_linear_scaled_q_columns_feed_the_prescaled_attention(lib):
    """The level-0 self-attention chain of the UNet as the executor runs it: fused LayerNorm + q/k/v projection whose first C columns leave multiplied
    by scale * log2(e) (fp32, before the one rounding), then ldiff_op_attention_prescaled -- against the plain chain (unscaled projection,
    ldiff_op_attention) and against torch.  The scaled columns themselves: y[:, :C] == f16(unscaled fp32 value * s) up to fp32 accumulation order."""
    Cc, heads, d, B, L = 320, 8, 40, 2, 256
    M = B * L
    g = torch.Generator().manual_seed(9)
    x = torch.randn((M, Cc), generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    w = torch.randn((3 * Cc, Cc), generator=g) / math.sqrt(Cc)
    xs = to_split(x).to(DEV)
    wd, gd, btd = w.to(torch.float16).contiguous().to(DEV), gamma.to(DEV), beta.to(DEV)
    s = (1.0 / math.sqrt(d)) * 1.4426950408889634

    def project(qcols, qscale):
        y = torch.full((M, 3 * Cc), float("nan"), dtype=torch.float16, device=DEV)
        _lib.check(lib.ldiff_op_ln_linear(xs.data_ptr(), 2 * Cc, Cc, M, Cc, gd.data_ptr(), btd.data_ptr(), 1e-5, wd.data_ptr(), 3 * Cc, 3 * Cc, None, 0, y.data_ptr(), 3 * Cc,
                                          qcols, qscale, sp()))
        return y
    plain, scaled = project(0, 1.0), project(Cc, s)
    torch.cuda.synchronize()
    a16 = r16(F.layer_norm(from_split(xs.cpu(), Cc), (Cc,), gamma, beta, 1e-5))
    proj = a16 @ r16(w).t()
    assert_close(scaled[:, :Cc], proj[:, :Cc] * s, "scaled q columns", rtol=2e-3, atol_rel=1.5e-3)
    assert torch.equal(scaled[:, Cc:], plain[:, Cc:]), "k / v columns must not change"
    o_pre = torch.full((B, L, Cc), float("nan"), dtype=torch.float16, device=DEV)
    o_ref = torch.empty_like(o_pre)
    bs, bp = scaled.data_ptr(), plain.data_ptr()
    _lib.check(lib.ldiff_op_attention_prescaled(bs, 3 * Cc, bs + 2 * Cc, 3 * Cc, bs + 4 * Cc, 3 * Cc, o_pre.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc, sp()))
    _lib.check(lib.ldiff_op_attention(bp, 3 * Cc, bp + 2 * Cc, 3 * Cc, bp + 4 * Cc, 3 * Cc, o_ref.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc,
                                      1.0 / math.sqrt(d), sp()))
    torch.cuda.synchronize()
    sh = lambda t: t.reshape(B, L, heads, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sh(proj[:, :Cc]), sh(proj[:, Cc:2 * Cc]), sh(proj[:, 2 * Cc:])).transpose(1, 2).reshape(B, L, Cc)
    assert_close(o_pre, want, "LN-linear(q scaled) -> prescaled attention vs torch", rtol=4e-3, atol_rel=4e-3)
    assert_close(o_pre, o_ref.float(), "prescaled chain vs plain chain", rtol=4e-3, atol_rel=4e-3)
    with pytest.raises(ValueError):
        project(100, s)        # not a multiple of the panel width
