"""In-kernel cycle stamps of the 8x16-tile conv3x3 kernel `conv3x3w_kernel` (diagnostic build: -DC3W_STAMPS, scripts/build_c3w_stamps.sh; never the product
library).  For wave 0 of workgroup 0: mean time per (slab, tap) step spent (1) issuing the step's LDS reads, the next step's weight DMA and, once per slab, the
next halo's loads; (2) in the MFMA groups with their operand waits and the GroupNorm transform of the next slab's chunk; (3) at the step's barrier, which also waits
for the weight DMA issued in (1).  s_memtime ticks are converted with the launch's own wall time.  usage: python scripts/conv_stamps_w.py"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import _lib

_lib.LIB_PATH = os.path.abspath(os.environ.get("LDIFF_LIB", "build/stamps_c3w/libldiff_hip.so"))
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
# name: (B, C1, C2, H, Cout)  -- GroupNorm + SiLU prologue, bias, fp16 output
SHAPES = {"L0 320->320 @64^2 (160-column kernel, one workgroup per CU)": (8, 320, 0, 64, 320),
          "L0 960->320 concat @64^2 (160-column kernel)": (8, 640, 320, 64, 320),
          "320->256 @64^2 (128-column kernel, two per CU, one round)": (8, 320, 0, 64, 256),
          "L1 640->640 @32^2 (128-column kernel, 320 workgroups)": (8, 640, 0, 32, 640),
          "L2 1280->1280 @16^2 (split-K 3)": (8, 1280, 0, 16, 1280)}
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, (B, C1, C2, H, Cout) in SHAPES.items():
    Cin = C1 + C2
    x = torch.randn((B, H, H, C1), device="cuda", dtype=torch.float16)
    x2 = torch.randn((B, H, H, C2), device="cuda", dtype=torch.float16) if C2 else None
    w = (torch.randn((Cout, 9 * Cin), device="cuda") / math.sqrt(9 * Cin)).to(torch.float16)
    y = torch.empty((B, H, H, Cout), device="cuda", dtype=torch.float16)
    bias = torch.randn(Cout, device="cuda")
    sc, sh = torch.rand((B, Cin), device="cuda") + 0.5, torch.randn((B, Cin), device="cuda") * 0.1
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = x.data_ptr(), C1, B, H, H, H, H, 3, 1, 1, 1
    if x2 is not None:
        a.x2, a.C2 = x2.data_ptr(), C2
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy = w.data_ptr(), Cout, Cout, bias.data_ptr(), y.data_ptr(), Cout
    a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), 1
    for _ in range(3):
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    buf = (C.c_ulonglong * 32)()
    assert raw.ldiff_debug_c3w_stamps(buf) == 0
    d = list(buf)
    steps = max(d[4], 1)
    loop = d[1] + d[2] + d[3]
    print(f"{name}: {us:.1f} us per launch ({2e-6 * B * H * H * Cout * 9 * Cin / us:.0f} TFLOP/s); workgroup 0, wave 0: {steps} steps, start -> end of the loop {d[0]} ticks;\n"
          f"    per step: issue {d[1] / steps:.0f}  mfma + transform {d[2] / steps:.0f}  barrier (waits for the weight DMA) {d[3] / steps:.0f}  ticks "
          f"= {100 * d[1] / loop:.0f} % / {100 * d[2] / loop:.0f} % / {100 * d[3] / loop:.0f} %; loop {100 * loop / max(d[0], 1):.0f} % of start -> end of loop\n"
          f"    per tap (x {steps // 9} slabs), MFMA segment: {[round(9 * v / steps) for v in d[8:17]]}  issue segment: {[round(9 * v / steps) for v in d[20:29]]}", flush=True)
