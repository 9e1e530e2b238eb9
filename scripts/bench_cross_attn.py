"""Diagnostic (GPU box): the UNet's cross-attention launches (K / V of the prompt, L_ctx = 6 or 77) at B = 8; us per call and algorithmic GB/s.
LDIFF_ATTN_HEADLOOP=0 restores one workgroup per (image, head, query tile)."""
import ctypes as C, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, heads, Lq, Lk, d) in ((8, 8, 4096, 6, 40), (8, 8, 4096, 77, 40), (8, 8, 1024, 6, 80), (8, 8, 256, 6, 160), (8, 8, 64, 6, 160)):
    Cc = heads * d
    q = torch.randn((B, Lq, Cc), device=DEV).half(); k = torch.randn((1, Lk, 2 * Cc), device=DEV).half()
    o = torch.empty_like(q)
    def f():
        _lib.check(lib.ldiff_op_attention(q.data_ptr(), Cc, k.data_ptr(), 2 * Cc, k.data_ptr() + 2 * Cc, 2 * Cc, o.data_ptr(), Cc, B, heads, Lq, Lk, d, Lq * Cc, 0, Lq * Cc,
                                          1.0 / math.sqrt(d), sp()))
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"cross-attention B={B} Lq={Lq} Lk={Lk} d={d}: {us:6.1f} us  {2 * q.numel() * 2 / us * 1e-3:7.1f} GB/s (Q read + O written)", flush=True)
