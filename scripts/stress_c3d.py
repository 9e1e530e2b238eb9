"""Race check of the dataflow conv3x3 kernel (diagnostic, GPU): the same launch repeated many times must reproduce its output and its fused
statistics bit for bit (producers and consumers meet only through progress words: a missing wait shows up as a now-and-then wrong tile).
usage: python scripts/stress_c3d.py [repeats]   (LDIFF_C3D_RUN / LDIFF_CONV3X3_DATAFLOW as for the library)"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
# (B, Cin, H, Cout, residual, statistics, channels of a folded 1x1 shortcut or 0)
for (B, Cin, H, Cout, res, stats, Cs) in [(8, 128, 512, 128, 1, 1, 0), (8, 128, 512, 128, 0, 1, 0), (8, 256, 256, 256, 1, 1, 0), (8, 512, 128, 512, 0, 0, 0), (3, 128, 256, 128, 1, 1, 0),
                                          (8, 128, 512, 128, 0, 1, 256), (8, 256, 256, 256, 0, 1, 512), (3, 128, 256, 128, 0, 0, 256)]:
    g = torch.Generator().manual_seed(B + Cin)
    x = torch.randn((B, H, H, Cin), generator=g).to(torch.float16).cuda()
    w = (torch.randn((Cout, 9 * Cin), generator=g) / math.sqrt(9 * Cin)).to(torch.float16).cuda()
    y = torch.empty((B, H, H, Cout), device="cuda", dtype=torch.float16)
    bias = torch.randn(Cout, generator=g).cuda()
    sc, sh = (torch.rand((B, Cin), generator=g) + 0.5).cuda(), (torch.randn((B, Cin), generator=g) * 0.1).cuda()
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = x.data_ptr(), Cin, B, H, H, H, H, 3, 1, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy = w.data_ptr(), Cout, Cout, bias.data_ptr(), y.data_ptr(), Cout
    a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), 1
    if res:
        r = torch.randn((B, H, H, Cout), generator=g).to(torch.float16).cuda()
        a.res, a.ld_res = r.data_ptr(), Cout
    if Cs:
        xs = torch.randn((B, H, H, Cs), generator=g).to(torch.float16).cuda()
        wsc = (torch.randn((Cout, Cs), generator=g) / math.sqrt(Cs)).to(torch.float16).cuda()
        bsc = torch.randn(Cout, generator=g).cuda()
        a.sc_x, a.sc_C, a.sc_ld, a.sc_w, a.sc_bias = xs.data_ptr(), Cs, Cs, wsc.data_ptr(), bsc.data_ptr()
    st = None
    if stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a))
        st = torch.empty((B, Cout, R, 2), device="cuda")
        a.stats = st.data_ptr()
    ref_y = ref_s = None
    bad = 0
    for it in range(reps):
        y.fill_(float("nan"))
        if st is not None:
            st.fill_(float("nan"))
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
        torch.cuda.synchronize()
        if ref_y is None:
            ref_y, ref_s = y.clone(), None if st is None else st.clone()
            assert torch.isfinite(ref_y).all() and (ref_s is None or torch.isfinite(ref_s).all())
        else:
            bad += int(not torch.equal(y, ref_y)) + int(st is not None and not torch.equal(st, ref_s))
    print(f"B{B} Cin{Cin} H{H} Cout{Cout} res{res} stats{stats} folded shortcut {Cs}: {reps} launches, {bad} differ from the first", flush=True)
    assert bad == 0
print("ok")
