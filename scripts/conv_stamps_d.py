"""In-kernel stamps of the dataflow conv3x3 kernel (diagnostic build: scripts/build_variant.sh stampsd -DC3D_STAMPS; never the product library).
usage: LDIFF_LIB=build/stampsd/libldiff_hip.so python scripts/conv_stamps_d.py [B Cin H Cout [res stats]] ..."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("LDIFF_LIB", "build/stampsd/libldiff_hip.so"))
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [(8, 128, 512, 128, 0, 0), (8, 128, 512, 128, 1, 1), (8, 256, 256, 256, 0, 0), (8, 512, 128, 512, 0, 0)]
for (B, Cin, H, Cout, res, stats) in cases:
    x = torch.randn((B, H, H, Cin), device="cuda", dtype=torch.float16)
    w = (torch.randn((Cout, 9 * Cin), device="cuda") / math.sqrt(9 * Cin)).to(torch.float16)
    y = torch.empty((B, H, H, Cout), device="cuda", dtype=torch.float16)
    bias = torch.randn(Cout, device="cuda")
    sc, sh = torch.rand((B, Cin), device="cuda") + 0.5, torch.randn((B, Cin), device="cuda") * 0.1
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = x.data_ptr(), Cin, B, H, H, H, H, 3, 1, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy = w.data_ptr(), Cout, Cout, bias.data_ptr(), y.data_ptr(), Cout
    a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), 1
    if res:
        r = torch.randn((B, H, H, Cout), device="cuda", dtype=torch.float16)
        a.res, a.ld_res = r.data_ptr(), Cout
    if stats:
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a))
        st = torch.empty((B, Cout, R, 2), device="cuda")
        a.stats = st.data_ptr()
    for _ in range(3):
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.ldiff_op_conv(C.byref(a), sp)); e1.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 48)()
    assert raw.ldiff_debug_c3d_stamps(out) == 0
    c, p = [out[i] for i in range(16)], [out[16 + i] for i in range(32)]
    us = e0.elapsed_time(e1) * 1e3
    units = max(c[4], 1); slabs = max(p[6], 1)
    print(f"B{B} Cin{Cin} H{H} Cout{Cout} res{res} stats{stats}: {us:.0f} us")
    print(f"  consumer w0: total {c[6]} ticks; first-slab wait {c[0]}; slab checks {c[2]}, polls while waiting {c[1]}; units {c[4]}: epilogue {c[3]/units:.0f} ticks each, hand-over to the next unit {c[5]/units:.0f};"
          f" loop share {(c[6]-c[0]-c[3]-c[5])/max(c[6],1):.3f}")
    print(f"    staging polls: producers' registers {c[7]}, consumers' last reads {c[8]}")
    print(f"  producer w4: total {p[7]} ticks; per slab ({slabs}): gate + issue {p[1]/slabs:.0f} ({p[5]} sleeps)  wait for the pieces {p[2]/slabs:.0f}  transform + publish {p[3]/slabs:.0f}"
          f"  store a tile + issue behind it {p[4]/slabs:.0f} (polls: staged {p[8]}, all producers read {p[9]})")
