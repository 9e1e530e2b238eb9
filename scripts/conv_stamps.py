"""In-kernel cycle stamps of the 16x16 ping-pong conv3x3 kernel (diagnostic build: -DC3P_STAMPS, never the product library).
build:  hipcc ... -DC3P_STAMPS into build/stamps/libldiff_hip.so (scripts/build_stamps.sh);  usage: python scripts/conv_stamps.py [shape...]
Prints, for wave 0 (group A) and wave 4 (group B) of workgroup 0, the mean cycles per step spent in the matrix segment, the
barrier after it, the load segment and the barrier after that."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import _lib

_lib.LIB_PATH = os.path.abspath(os.environ.get("LDIFF_LIB", "build/stamps/libldiff_hip.so"))
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
SHAPES = {"one_wg_128_128": (1, 128, 64, 128, 0), "128_128_512": (8, 128, 512, 128, 0), "128_128_512_gn": (8, 128, 512, 128, 1), "512_512_128_gn": (8, 512, 128, 512, 1), "256_256_256": (8, 256, 256, 256, 0)}
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, (B, Cin, H, Cout, gn) in SHAPES.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    x = torch.randn((B, H, H, Cin), device="cuda", dtype=torch.float16)
    w = (torch.randn((Cout, 9 * Cin), device="cuda") / math.sqrt(9 * Cin)).to(torch.float16)
    y = torch.empty((B, H, H, Cout), device="cuda", dtype=torch.float16)
    bias = torch.randn(Cout, device="cuda")
    sc, sh = torch.rand((B, Cin), device="cuda") + 0.5, torch.randn((B, Cin), device="cuda") * 0.1
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = x.data_ptr(), Cin, B, H, H, H, H, 3, 1, 1, 1
    a.w, a.N, a.Nrows, a.bias, a.y, a.ldy = w.data_ptr(), Cout, Cout, bias.data_ptr(), y.data_ptr(), Cout
    if gn:
        a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), 1
    if os.environ.get('LDIFF_BENCH_STATS'):   # with fused GroupNorm statistics (epilogue configuration 4)
        R = lib.ldiff_op_conv_stats_blocks(C.byref(a))
        st = torch.empty((B, Cout, R, 2), device='cuda')
        a.stats = st.data_ptr()
    for _ in range(3):
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.ldiff_op_conv(C.byref(a), sp)); e1.record(); torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert raw.ldiff_debug_c3p_stamps(out) == 0
    print(f"{name}: {e0.elapsed_time(e1)*1e3:.0f} us")
    for g in range(2):
        d = [out[g * 16 + i] for i in range(16)]
        n = max(d[4], 1)
        print(f"  group {'AB'[g]}: steps {d[4]}  matrix {d[0]/n:.0f}  barrier {d[1]/n:.0f} (of which vmcnt wait {d[14]/n:.0f})  load {d[2]/n:.0f}  barrier {d[3]/n:.0f}  sum {(d[0]+d[1]+d[2]+d[3])/n:.0f} cycles/step;"
              f"  load seg at tap 0: {d[5]/max(n/9,1):.0f}; tap 3: {d[13]/max(n/9,1):.0f}; epilogue load seg: {d[6]/max(d[7],1):.0f} (x{d[7]}) = before {d[11]/max(d[7],1):.0f} + stores {d[8]/max(d[7],1):.0f} + init {d[9]/max(d[7],1):.0f} + rest {d[10]/max(d[7],1):.0f}")
