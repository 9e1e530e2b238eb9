"""Diagnostic for the dataflow conv3x3 kernel: run one case through ldiff_op_conv, compare with the CPU reference and print where the
errors sit (image, 16x16 tile, channel tile of 128, wave quadrant = (row half, channel half of 64)).
usage: python scripts/debug_c3d.py B Cin H W Cout [res] [temb]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
from tests.test_gpu_kernels import run_conv

B, Cin, H, W, Cout = [int(a) for a in sys.argv[1:6]]
use_res = len(sys.argv) > 6 and sys.argv[6] == "1"
use_temb = len(sys.argv) > 7 and sys.argv[7] == "1"
lib = _lib.load()
g = torch.Generator().manual_seed(5)
x = torch.randn((B, Cin, H, W), generator=g)
w = torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(Cin * 9)
bias = torch.randn(Cout, generator=g) * 0.1
gn = (1.0 + 0.2 * torch.randn((B, Cin), generator=g), 0.2 * torch.randn((B, Cin), generator=g))
temb = torch.randn((B, Cout), generator=g) * 0.3 if use_temb else None
res = torch.randn((B, Cout, H, W), generator=g) if use_res else None
for it in range(3):
    got, ref = run_conv(lib, x, w, bias, None, 1, (1, 1), 0, gn, 1, temb, res, False, False)
    err = (got - ref).abs()
    err[~torch.isfinite(got)] = 1e9
    tol = 2e-3 * ref.abs().max() + 2e-3 * ref.abs()
    bad = err > tol
    print(f"launch {it}: bad {int(bad.sum())}/{bad.numel()}  nan {int((~torch.isfinite(got)).sum())}  max finite err {err[err < 1e8].max():.3e}")
    if bad.any():
        # [B, Cout, H, W] -> per (b, ty, tx, nt, wave_m, wave_n)
        e = bad.reshape(B, Cout // 128, 2, 64, H // 16, 2, 8, W // 16, 16).float()
        per = e.mean(dim=(3, 6, 8))   # [B, nt, wave_n, ty, wave_m, tx]
        idx = (per > 0).nonzero()
        print("  bad (b, ntile, wave_n, ty, wave_m, tx) count:", len(idx), " of ", per.numel())
        for r in idx[:40].tolist():
            print("   ", r, f"{per[tuple(r)]:.3f}")
        # which pixel rows / cols / channels inside a bad quadrant
        b, nt, wn, ty, wm, tx = idx[0].tolist()
        q = err[b, nt * 128 + wn * 64: nt * 128 + wn * 64 + 64, ty * 16 + wm * 8: ty * 16 + wm * 8 + 8, tx * 16: tx * 16 + 16]
        print("  first bad quadrant: err by channel tile of 16:", [f"{q[a*16:(a+1)*16].clamp(max=10).mean():.3f}" for a in range(4)])
        print("  by pixel row:", [f"{q[:, m].clamp(max=10).mean():.3f}" for m in range(8)])
        print("  by pixel col:", [f"{q[:, :, c].clamp(max=10).mean():.3f}" for c in range(16)])
