"""Race check (diagnostic, GPU) for the producer / consumer GEMM (kernels_gemm_df.hip: LDS-DMA ring with counted vmcnt, progress words, slice hand-over,
no barrier): several shapes x epilogues x unit shapes, launched many times while a second stream keeps the chip busy with conv3x3 launches (other
landing times for the DMA pieces, uneven CU availability, other dispatch orders), back to back with itself, every result compared bit for bit with
the result computed alone AND with the LDS-DMA GEMM's.  usage: python scripts/stress_gemm_df.py [iters]"""
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
# (M, K, N, epilogue, plan)
CASES = [(32768, 320, 320, "split_res_out", 133), (32768, 320, 960, "plain", 133), (8192 + 77, 640, 1920, "plain", 132), (8192, 640, 5120, "geglu", 132),
         (8192, 2560, 640, "split_res_out", 69), (2048, 1280, 10240, "geglu", 133), (4096, 128, 256, "split_res_out", 130), (32768, 640, 320, "split_res_out+stats", 133)]
for M, K, N, epi, plan in CASES:
    stats = epi.endswith("+stats"); epi = epi.replace("+stats", "")
    geglu = epi == "geglu"
    x = torch.randn((M, K), generator=g).half().to(DEV)
    w = (torch.randn((N, K), generator=g) / math.sqrt(K)).half().to(DEV)
    b = (torch.randn(N, generator=g) * 0.2).to(DEV)
    Nout = N // 2 if geglu else N
    split = epi == "split_res_out"
    a = _lib.ConvArgs()
    Bimg = 8 if stats else 1
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride = x.data_ptr(), K, Bimg, 1, M // Bimg, 1, M // Bimg, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.geglu = w.data_ptr(), N, N, b.data_ptr(), int(geglu)
    if split:
        res = torch.randn((M, 2 * N), generator=g).half().to(DEV)
        a.res, a.ld_res, a.res_lo = res.data_ptr(), 2 * N, N
    cols = 2 * Nout if split else Nout
    R = M // Bimg // 32

    def run(y, st, s, df):
        a.y, a.ldy, a.y_lo, a.gemm_df = y.data_ptr(), cols, Nout if split else 0, df
        if stats:
            a.stats = st.data_ptr()
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp(s)))
    ref, ref_st = torch.empty((M, cols), dtype=torch.float16, device=DEV), torch.empty((Bimg, N, R, 2), device=DEV)
    dma, dma_st = torch.empty_like(ref), torch.empty_like(ref_st)
    torch.cuda.synchronize()
    run(ref, ref_st, sa, plan); run(dma, dma_st, sa, -1); torch.cuda.synchronize()
    assert torch.equal(ref, dma) and (not stats or torch.equal(ref_st, dma_st)), "dataflow and LDS-DMA GEMM disagree"
    for it in range(iters):
        y = torch.full((M, cols), float("nan"), dtype=torch.float16, device=DEV)
        st = torch.full((Bimg, N, R, 2), float("nan"), device=DEV)
        torch.cuda.synchronize()
        for k in range(1 + it % 3):
            _lib.check(lib.ldiff_op_conv(C.byref(cb), sp(sb)))
        run(y, st, sa, plan)
        if it % 2:
            run(y, st, sa, plan)                                    # back to back: the next launch's prologue under this one's tail
        torch.cuda.synchronize()
        if not torch.equal(y, ref) or (stats and not torch.equal(st, ref_st)):
            bad += 1
            d = (y.float() - ref.float()).abs()
            print(f"MISMATCH M={M} K={K} N={N} {epi} iteration {it}: {int((d > 0).sum())} elements, max {d.max().item():.3e}, nan {int(torch.isnan(y).sum())}", flush=True)
    print(f"M={M} K={K} N={N} {epi}{'+stats' if stats else ''} plan {plan >> 4}x{plan & 15}: {iters} launches under load, mismatches so far {bad}", flush=True)
print("ok" if bad == 0 else f"FAILED: {bad} mismatching launches")
sys.exit(1 if bad else 0)
