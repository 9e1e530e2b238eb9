"""one-hot weight probes for the dataflow conv3x3 kernel"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
from tests.test_gpu_kernels import run_conv
lib = _lib.load()
B, Cin, H, W, Cout = 1, 64, 256, 256, 128
g = torch.Generator().manual_seed(5)
x = torch.randint(-8, 9, (B, Cin, H, W), generator=g).float()
gn = (torch.ones((B, Cin)), torch.zeros((B, Cin)))
for (c0, ky, kx) in [(0, 1, 1), (5, 1, 1), (37, 1, 1), (0, 0, 0), (9, 2, 1), (63, 1, 2)]:
    w = torch.zeros((Cout, Cin, 3, 3))
    w[:, c0, ky, kx] = 1.0
    got, ref = run_conv(lib, x, w, None, None, 1, (1, 1), 0, gn, 0, None, None, False, False)
    bad = (got - ref).abs() > 1e-3
    print(f"c0={c0} tap=({ky},{kx}): bad {int(bad.sum())}/{bad.numel()}")
    if bad.any():
        # try to explain got[co, y, x] as x[c', y+dy, x+dx]
        patch = got[0, 0, 32:40, 32:48]
        best = None
        for c in range(Cin):
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    cand = x[0, c, 32 + dy:40 + dy, 32 + dx:48 + dx]
                    if torch.equal(cand, patch):
                        best = (c, dy, dx)
        print("   out channel 0 equals x[c, y+dy, x+dx] for", best, " expected", (c0, ky - 1, kx - 1))
        print("   got[0,0,32,32:40]", got[0, 0, 32, 32:40].tolist(), " ref", ref[0, 0, 32, 32:40].tolist())
        print("   per out-channel bad fraction:", [round(float(bad[0, c].float().mean()), 2) for c in range(0, 128, 8)])
print("--- random weights, GN / SiLU variants")
w = torch.randn((Cout, Cin, 3, 3), generator=g) / math.sqrt(Cin * 9)
xr = torch.randn((B, Cin, H, W), generator=g)
sc, sh = 1.0 + 0.2 * torch.randn((B, Cin), generator=g), 0.2 * torch.randn((B, Cin), generator=g)
for name, gnv, silu in [("identity nosilu", gn, 0), ("identity silu", gn, 1), ("scale only nosilu", (sc, torch.zeros((B, Cin))), 0), ("shift only nosilu", (torch.ones((B, Cin)), sh), 0),
                        ("full nosilu", (sc, sh), 0), ("full silu", (sc, sh), 1)]:
    got, ref = run_conv(lib, xr, w, None, None, 1, (1, 1), 0, gnv, silu, None, None, False, False)
    err = (got - ref).abs()
    print(f"{name:22s} max err {err.max():.4e}  mean err {err.mean():.4e}  max|ref| {ref.abs().max():.3f}")
