"""Diagnostic (GPU box): LayerNorm + linear as two launches (ldiff_op_layernorm + ldiff_op_conv) against the fused activation-stationary
kernel (ldiff_op_ln_linear) on the UNet's C = 320 shapes at B = 8 (M = 32768), interleaved rounds in one process; us per call."""
import ctypes as C, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, Cc = 32768, 320
g = torch.Generator().manual_seed(0)
x = torch.randn((M, Cc), generator=g) * 2
hi = x.to(torch.float16); lo = (x - hi.float()).to(torch.float16)
xs = torch.cat([hi, lo], -1).contiguous().to(DEV)
gamma, beta = (1 + 0.1 * torch.randn(Cc, generator=g)).to(DEV), (0.1 * torch.randn(Cc, generator=g)).to(DEV)
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, N, geglu in (("qkv 320->960", 960, 0), ("q2 320->320", 320, 0), ("ff1 320->2560 geglu", 2560, 1)):
    w = (torch.randn((N, Cc), generator=g) / math.sqrt(Cc)).to(torch.float16).to(DEV)
    b = (torch.randn(N, generator=g) * 0.2).to(DEV)
    Nout = N // 2 if geglu else N
    n = torch.empty((M, Cc), dtype=torch.float16, device=DEV)
    y = torch.empty((M, Nout), dtype=torch.float16, device=DEV)
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride = n.data_ptr(), Cc, 1, 1, M, 1, M, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy, a.geglu = w.data_ptr(), N, N, b.data_ptr(), y.data_ptr(), Nout, geglu
    def two():
        _lib.check(lib.ldiff_op_layernorm(xs.data_ptr(), 2 * Cc, Cc, n.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-5, sp()))
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp()))
    def ln_only():
        _lib.check(lib.ldiff_op_layernorm(xs.data_ptr(), 2 * Cc, Cc, n.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-5, sp()))
    def one():
        _lib.check(lib.ldiff_op_ln_linear(xs.data_ptr(), 2 * Cc, Cc, M, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), N, N, b.data_ptr(), geglu, y.data_ptr(), Nout, 0, 1.0, sp()))
    res = []
    for r in range(3):
        res.append((timeit(two), timeit(ln_only), timeit(one)))
    t2, tl, t1 = (min(v[i] for v in res) for i in range(3))
    fl = 2.0 * M * N * Cc
    print(f"{name:22s} two launches {t2:7.1f} us (layernorm alone {tl:5.1f})   fused {t1:7.1f} us = {fl / t1 * 1e-6:7.1f} TFLOP/s   ratio {t2 / t1:.2f}x", flush=True)
