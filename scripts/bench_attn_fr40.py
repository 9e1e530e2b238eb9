"""The UNet's level-0 self-attention (8 heads, d = 40, fused q/k/v rows of 960) at B = 8, 4,096 tokens, and at the 1024^2 ROI shape (B = 2,
16,384 tokens): the fixed-reference kernel (LDIFF_ATTN_FIXREF=1, default) against the generic one (=0), each in its own process.
usage: python scripts/bench_attn_fr40.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import math, sys, torch
sys.path.insert(0, %r)
from ldiffusion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for B, L in ((8, 4096), (2, 16384)):
    heads, d = 8, 40
    Cc = heads * d
    qkv = torch.randn((B, L, 3 * Cc), generator=torch.Generator().manual_seed(1)).to(torch.float16).to(dev)
    o = torch.empty((B, L, Cc), dtype=torch.float16, device=dev)
    base = qkv.data_ptr()
    def run():
        _lib.check(lib.ldiff_op_attention(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc,
                                          1.0 / math.sqrt(d), _lib.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 4.0 * B * heads * L * L * d
    print(f"  B={B} L={L}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s   checksum {float(o.float().abs().mean()):.6f}", flush=True)
''' % ROOT
for mode in ("0", "1"):
    print(f"LDIFF_ATTN_FIXREF={mode}", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LDIFF_ATTN_FIXREF=mode), check=True)
