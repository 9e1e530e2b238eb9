"""Race check (diagnostic, GPU) of the tap-folded conv_out kernel (conv3x3nt_kernel): its workgroups hand the tap products of a tile through a
double-buffered LDS array with ONE barrier per tile -- a missing wait would show up as a now-and-then wrong pixel.  Many launches of the VAE
shape (16 tiles per workgroup) and of a ragged map, with the uint8 / luma tail, every output compared bit for bit with the first launch's.
usage: python scripts/stress_conv_out.py [launches]"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import _lib

lib = _lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, W, silu) in [(8, 512, 512, 1), (3, 200, 216, 1), (2, 72, 40, 0)]:
    g = torch.Generator().manual_seed(B + H)
    Cin, Cout, Nrows = 128, 3, 16
    x = torch.randn((B, H, W, Cin), generator=g).to(torch.float16).cuda()
    w = torch.zeros((Nrows, 9 * Cin), dtype=torch.float16)
    w[:Cout] = (torch.randn((Cout, 9 * Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16)
    w = w.cuda()
    bias = torch.zeros(Nrows); bias[:Cout] = torch.randn(Cout, generator=g) * 0.1; bias = bias.cuda()
    sc, sh = (torch.rand((B, Cin), generator=g) + 0.5).cuda(), (torch.randn((B, Cin), generator=g) * 0.1).cuda()
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = x.data_ptr(), Cin, B, H, W, H, W, 3, 1, 1, 1
    a.w, a.N, a.Nrows, a.n_real, a.bias = w.data_ptr(), 4, Nrows, Cout, bias.data_ptr()
    a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), silu
    y = torch.empty((B, H, W, 4), device="cuda")
    a.y, a.ldy, a.out_f32 = y.data_ptr(), 4, 1
    first, bad = None, 0
    for it in range(reps):
        y.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
        torch.cuda.synchronize()
        if first is None:
            first = y.clone()
            assert torch.isfinite(first).all()
        elif not torch.equal(y, first):
            bad += 1
    print(f"B{B} {H}x{W} silu{silu}: {reps} launches, {bad} differ from the first", flush=True)
    assert bad == 0
print("ok")
