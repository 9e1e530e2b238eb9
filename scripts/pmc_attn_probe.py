"""Five launches of the UNet's level-0 self-attention (B = 8, 8 heads, d = 40, 4,096 tokens, fused q/k/v) for a rocprofv3 --pmc pass.
usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 scripts/pmc_attn_probe.py <0|1>   (0: generic kernel, 1: fixed-reference kernel)"""
import math, os, sys
os.environ["LDIFF_ATTN_FIXREF"] = sys.argv[1] if len(sys.argv) > 1 else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
B, heads, L, d = 8, 8, 4096, 40
Cc = heads * d
qkv = torch.randn((B, L, 3 * Cc), generator=torch.Generator().manual_seed(1)).half().to("cuda:0")
o = torch.empty((B, L, Cc), dtype=torch.float16, device="cuda:0")
base = qkv.data_ptr()
for _ in range(5):
    _lib.check(lib.ldiff_op_attention(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, o.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc,
                                      1.0 / math.sqrt(d), _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
