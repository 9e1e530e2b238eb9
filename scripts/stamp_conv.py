"""Diagnostic: where a conv3x3 tap step spends its cycles (s_memtime stamps; cdna_hip_programming.md 'In-kernel stamps').
Builds a SEPARATE library with -DLDIFF_STAMPS (gpurun_out/libldiff_stamps.so, never the product library), runs one conv
shape through ldiff_op_conv and prints the per-phase shares of wave cycles, median over waves.
usage: python scripts/stamp_conv.py build            (on the build host: hipcc)
       python scripts/stamp_conv.py run [shape ...]  (on the GPU box)
"""
import ctypes as C
import glob
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "ldiffusion_amd", "libldiff_stamps.so")

SHAPES = {  # name: (B, Cin, H, W, Cout, gn)
    "vae512_128_128_gn": (8, 128, 512, 512, 128, 1),
    "vae512_128_128_nogn": (8, 128, 512, 512, 128, 0),
    "vae256_256_256_gn": (8, 256, 256, 256, 256, 1),
    "vae128_512_512_gn": (8, 512, 128, 128, 512, 1),
    "vae128_512_512_nogn": (8, 512, 128, 128, 512, 0),
}


def build():
    srcs = sorted(glob.glob(os.path.join(ROOT, "ldiffusion_amd", "csrc", "*.hip")))
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-comment", "-DLDIFF_STAMPS", "-o", LIB] + srcs
    subprocess.run(cmd, check=True, cwd=ROOT)
    print("built", LIB)


def run(names):
    import numpy as np
    import torch
    from ldiffusion_amd import _lib
    _lib.LIB_PATH = LIB
    lib = _lib.load()
    lib.ldiff_debug_stamps.restype = C.c_int
    lib.ldiff_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name in names or list(SHAPES):
        B, Cin, H, W, Cout, gn = SHAPES[name]
        x = torch.randn((B, H, W, Cin), device="cuda", dtype=torch.float16)
        w = (torch.randn((Cout, 9 * Cin), device="cuda") / math.sqrt(9 * Cin)).to(torch.float16)
        y = torch.empty((B, H, W, Cout), device="cuda", dtype=torch.float16)
        bias = torch.randn(Cout, device="cuda")
        sc, sh = torch.rand((B, Cin), device="cuda") + 0.5, torch.randn((B, Cin), device="cuda") * 0.1
        a = _lib.ConvArgs()
        a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l = x.data_ptr(), Cin, B, H, W, H, W, 3, 1, 1, 1
        a.w, a.N, a.Nrows, a.bias, a.y, a.ldy = w.data_ptr(), Cout, Cout, bias.data_ptr(), y.data_ptr(), Cout
        if gn:
            a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), 1
        for _ in range(300):   # hold the load long enough for the clock to settle
            _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
        e1.record()
        torch.cuda.synchronize()
        n = 4096 * 4 * 8
        buf = np.zeros(n, dtype=np.uint64)
        rc = lib.ldiff_debug_stamps(buf.ctypes.data, n)
        assert rc == 0, rc
        raw = buf.reshape(4096 * 4, 8)
        raw = raw[raw[:, 7] > 0]
        rt = (raw[:, 7] >> np.uint64(16)).astype(np.float64)       # s_memrealtime ticks (100 MHz)
        s = raw.astype(np.float64)
        s[:, 7] = (raw[:, 7] & np.uint64(0xFFFF)).astype(np.float64)
        clock_ghz = np.median(s[:, 6] / rt) * 0.1
        med = np.median(s, axis=0)
        tot = med[6]
        labels = ["prologue", "issue(dma+halo ld)", "ds_read+mfma", "store_halo(gn)", "wait+barrier", "epilogue", "total", "steps"]
        mfma_min = med[7] * 32 * 16   # 32 MFMA of 16 cycles per step: the floor of the ds_read+mfma segment at 1 wave/SIMD
        print(f"{name}: {e0.elapsed_time(e1)*1e3:.0f} us (stamped build), in-kernel clock {clock_ghz:.2f} GHz, waves={len(s)}, steps={med[7]:.0f}, mfma floor {mfma_min:.0f} cyc = {mfma_min/tot:.1%} of wave time")
        for k in range(7):
            print(f"    {labels[k]:22s} {med[k]:10.0f} cyc  {med[k]/tot:6.1%}   per step {med[k]/med[7]:8.0f}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2:])
