"""Race check (diagnostic, GPU) for the LayerNorm-fused GEMM (kernels_gemm_ast.hip: weight ring by LDS-DMA, counted vmcnt, one barrier per panel):
the three UNet shapes at B = 8, launched many times while a second stream keeps the chip busy with conv3x3 launches (other landing times for
the DMA pieces, other dispatch orders), every result compared bit for bit with the result computed alone.  usage: python scripts/stress_lngemm.py [iters]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator().manual_seed(3)
Cc = 320
sp = lambda s: C.c_void_p(s.cuda_stream)
# background load: a 128 -> 128 conv3x3 on 2 x 512 x 512 (the 8x16 halo-tile kernel), on its own stream
xb = torch.randn((2, 512, 512, 128), generator=g).half().to(DEV)
wb = (torch.randn((128, 9 * 128), generator=g) / 34).half().to(DEV)
yb = torch.empty((2, 512, 512, 128), dtype=torch.float16, device=DEV)
cb = _lib.ConvArgs()
cb.x, cb.C1, cb.B, cb.Hin, cb.Win, cb.Hout, cb.Wout, cb.ks, cb.stride, cb.pad_t, cb.pad_l = xb.data_ptr(), 128, 2, 512, 512, 512, 512, 3, 1, 1, 1
cb.w, cb.N, cb.Nrows, cb.y, cb.ldy = wb.data_ptr(), 128, 128, yb.data_ptr(), 128
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
for M in (32768, 8192 + 77):
    x = torch.randn((M, Cc), generator=g) * 2
    hi = x.half(); xs = torch.cat([hi, (x - hi.float()).half()], -1).contiguous().to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(Cc, generator=g)).to(DEV), (0.1 * torch.randn(Cc, generator=g)).to(DEV)
    for N, geglu in ((960, 0), (320, 0), (2560, 1)):
        w = (torch.randn((N, Cc), generator=g) / math.sqrt(Cc)).half().to(DEV)
        b = (torch.randn(N, generator=g) * 0.2).to(DEV)
        Nout = N // 2 if geglu else N
        def run(y, s):
            _lib.check(lib.ldiff_op_ln_linear(xs.data_ptr(), 2 * Cc, Cc, M, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), N, N, b.data_ptr(), geglu, y.data_ptr(), Nout, 0, 1.0, sp(s)))
        ref = torch.empty((M, Nout), dtype=torch.float16, device=DEV)
        torch.cuda.synchronize()
        run(ref, sa); torch.cuda.synchronize()
        for it in range(iters):
            y = torch.full((M, Nout), float("nan"), dtype=torch.float16, device=DEV)
            torch.cuda.synchronize()
            for k in range(1 + it % 3):
                _lib.check(lib.ldiff_op_conv(C.byref(cb), sp(sb)))
            run(y, sa)
            if it % 2:
                run(y, sa)                                    # back to back: the next launch's prologue under this one's tail
            torch.cuda.synchronize()
            if not torch.equal(y, ref):
                bad += 1
                d = (y.float() - ref.float()).abs()
                print(f"MISMATCH M={M} N={N} geglu={geglu} iteration {it}: {int((d > 0).sum())} elements, max {d.max().item():.3e}, nan {int(torch.isnan(y).sum())}", flush=True)
        print(f"M={M} N={N} geglu={geglu}: {iters} launches under load, mismatches so far {bad}", flush=True)
print("ok" if bad == 0 else f"FAILED: {bad} mismatching launches")
sys.exit(1 if bad else 0)
