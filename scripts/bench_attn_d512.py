"""The VAE mid-block attention (one head, d = 512) at the bench shape (B = 8, 4,096 tokens, fused q/k/v rows of 1,536) and the 1024^2 ROI shape
(B = 2, 16,384 tokens): the 128-query fixed-reference kernel (LDIFF_ATTN_D512=1, default) against the d-split kernel (=0), each in its own
process (the switch is read once).  usage: python scripts/bench_attn_d512.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import math, sys, time, torch
sys.path.insert(0, %r)
from ldiffusion_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
for B, L in ((8, 4096), (2, 16384)):
    d = 512
    qkv = (torch.randn((B, L, 3 * d), generator=torch.Generator().manual_seed(1)) ).to(torch.float16).to(dev)
    o = torch.empty((B, L, d), dtype=torch.float16, device=dev)
    base = qkv.data_ptr()
    def run():
        _lib.check(lib.ldiff_op_attention(base, 3 * d, base + 2 * d, 3 * d, base + 4 * d, 3 * d, o.data_ptr(), d, B, 1, L, L, d, L * 3 * d, L * 3 * d, L * d,
                                          1.0 / math.sqrt(d), _lib.stream_ptr()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 4.0 * B * L * L * d
    print(f"  B={B} L={L}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s   checksum {float(o.float().abs().mean()):.6f}", flush=True)
''' % ROOT
for mode in ("0", "1"):
    print(f"LDIFF_ATTN_D512={mode}", flush=True)
    env = dict(os.environ, LDIFF_ATTN_D512=mode)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
