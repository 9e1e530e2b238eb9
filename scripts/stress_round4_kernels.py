"""Race check (diagnostic, GPU) for the three kernels added late in round 4 -- attn_fr40_kernel, attn_d512_kernel (raw barriers, double-buffered
LDS-DMA, hand-counted waits) and the fp8-lo path of the ping-pong conv kernel -- each launched many times while a second stream keeps the chip busy
with conv3x3 launches, every result compared bit for bit with the result computed alone.  usage: python scripts/stress_round4_kernels.py [iters]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(5)
sp = lambda s: C.c_void_p(s.cuda_stream)
xb = torch.randn((2, 512, 512, 128), generator=g).half().to(DEV)
wb = (torch.randn((128, 9 * 128), generator=g) / 34).half().to(DEV)
yb = torch.empty((2, 512, 512, 128), dtype=torch.float16, device=DEV)
cb = _lib.ConvArgs()
cb.x, cb.C1, cb.B, cb.Hin, cb.Win, cb.Hout, cb.Wout, cb.ks, cb.stride, cb.pad_t, cb.pad_l = xb.data_ptr(), 128, 2, 512, 512, 512, 512, 3, 1, 1, 1
cb.w, cb.N, cb.Nrows, cb.y, cb.ldy = wb.data_ptr(), 128, 128, yb.data_ptr(), 128
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0


def hammer(name, run, out_like):
    global bad
    ref = torch.empty_like(out_like)
    torch.cuda.synchronize()
    run(ref, sa); torch.cuda.synchronize()
    for it in range(iters):
        y = torch.full_like(out_like, float("nan"))
        torch.cuda.synchronize()
        for k in range(1 + it % 3):
            _lib.check(lib.ldiff_op_conv(C.byref(cb), sp(sb)))
        run(y, sa)
        if it % 2:
            run(y, sa)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            d = (y.float() - ref.float()).abs()
            print(f"MISMATCH {name} iteration {it}: {int((d > 0).sum())} elements, max {d.max().item():.3e}, nan {int(torch.isnan(y.float()).sum())}", flush=True)
    print(f"{name}: {iters} launches under load, mismatches so far {bad}", flush=True)


# ---- attention: d = 40 (8 heads, fused q/k/v) incl. a late spike that forces the repeat pass; d = 512 (one head) ----
for name, B, heads, L, d, spike in (("attn d=40 L=4096", 8, 8, 4096, 40, False), ("attn d=40 L=1000 repeat pass", 2, 8, 1000, 40, True),
                                     ("attn d=512 L=4096", 8, 1, 4096, 512, False), ("attn d=512 L=777 repeat pass", 2, 1, 777, 512, True)):
    Cc = heads * d
    qkv = torch.randn((B, L, 3 * Cc), generator=g)
    if spike:
        qkv[0, L - 3, Cc:Cc + d] = (12.0 if d == 40 else 2.0) * qkv[0, 5, :d]
    qkv = qkv.half().to(DEV)
    base = qkv.data_ptr()
    def run(y, s, base=base, Cc=Cc, B=B, heads=heads, L=L, d=d):
        _lib.check(lib.ldiff_op_attention(base, 3 * Cc, base + 2 * Cc, 3 * Cc, base + 4 * Cc, 3 * Cc, y.data_ptr(), Cc, B, heads, L, L, d, L * 3 * Cc, L * 3 * Cc, L * Cc,
                                          1.0 / math.sqrt(d), sp(s)))
    hammer(name, run, torch.empty((B, L, Cc), dtype=torch.float16, device=DEV))

# ---- ping-pong conv with an fp8 lo half (split output, statistics, split residual) ----
B, H, W, Cin, Cout = 8, 128, 128, 128, 128
x = torch.randn((B, H, W, Cin), generator=g)
x32 = torch.cat([x.half(), (x - x.half().float()).half()], -1).contiguous().to(DEV)
ones, zeros = torch.ones((B, Cin), device=DEV), torch.zeros((B, Cin), device=DEV)
xq = torch.empty((B, H, W, 3 * Cin), dtype=torch.uint8, device=DEV)
_lib.check(lib.ldiff_op_norm_apply_lo8(x32.data_ptr(), Cin, 2 * Cin, Cin, B, H * W, ones.data_ptr(), zeros.data_ptr(), 0, xq.data_ptr(), sp(sa)))
w = (torch.randn((Cout, 9 * Cin), generator=g) / math.sqrt(9 * Cin)).half().to(DEV)
wq = torch.empty((Cout, 9, 3 * Cin), dtype=torch.uint8, device=DEV)
wsc = torch.zeros(4, dtype=torch.int32, device=DEV)
_lib.check(lib.ldiff_op_lo8_weights(w.data_ptr(), wq.data_ptr(), wsc.data_ptr(), Cout, 9, Cin, sp(sa)))
r = torch.randn((B, H, W, Cout), generator=g)
rd = torch.cat([r.half(), (r - r.half().float()).half()], -1).contiguous().to(DEV)
bias = torch.randn(Cout, generator=g).to(DEV)
a = _lib.ConvArgs()
a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = xq.data_ptr(), Cin + Cin // 2, B, H, W, H, W, 3, 1, 1, 1
a.w, a.N, a.Nrows, a.bias, a.lo8_slab0, a.lo8_scale = wq.data_ptr(), Cout, Cout, bias.data_ptr(), Cin // 64, wsc.data_ptr()
a.res, a.ld_res, a.res_lo = rd.data_ptr(), 2 * Cout, Cout
ytmp = torch.empty((B, H, W, 2 * Cout), dtype=torch.float16, device=DEV)
a.y, a.ldy, a.y_lo = ytmp.data_ptr(), 2 * Cout, Cout
R = lib.ldiff_op_conv_stats_blocks(C.byref(a))
assert R > 0
st = torch.empty((B, Cout, R, 2), device=DEV)
a.stats = st.data_ptr()
torch.cuda.synchronize()
def run_conv(y, s):
    a.y, a.ldy, a.y_lo = y.data_ptr(), 2 * Cout, Cout
    _lib.check(lib.ldiff_op_conv(C.byref(a), sp(s)))
hammer("conv3x3 ping-pong, fp8 lo half", run_conv, torch.empty((B, H, W, 2 * Cout), dtype=torch.float16, device=DEV))
print("ok" if bad == 0 else f"FAILED: {bad} mismatching launches")
sys.exit(1 if bad else 0)
