"""Diagnostic (GPU box): level-0 / level-1 self-attention of the UNet at B = 8, plain (ldiff_op_attention) against the prescaled form, us per call."""
import ctypes as C, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, heads, L, d) in ((8, 8, 4096, 40), (8, 8, 1024, 80), (2, 8, 16384, 40)):
    Cc = heads * d
    qkv = torch.randn((B, L, 3 * Cc), device=DEV).half()
    o = torch.empty((B, L, Cc), dtype=torch.float16, device=DEV)
    base = qkv.data_ptr()
    def plain():
        _lib.check(lib.ldiff_op_attention(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc, 1.0 / math.sqrt(d), sp()))
    def pre():
        _lib.check(lib.ldiff_op_attention_prescaled(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc, sp()))
    def t(f, n=20):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    r = [(t(plain), t(pre)) for _ in range(3)]
    print(f"self-attention B={B} L={L} d={d}: plain {min(a for a, _ in r):7.1f} us   prescaled {min(b for _, b in r):7.1f} us", flush=True)
