"""Micro-benchmark of the contraction kernels through the C ABI (ldiff_op_conv / ldiff_op_attention) at the shapes the
B=8, 512x512 sampler launches.  Prints TFLOP/s per shape (HIP events on the launch stream, random data).
usage: python scripts/bench_conv.py [filter-substring] [--iters N]
"""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import _lib

DEV = "cuda:0"
# name: (B, C1, C2, Hin, Win, Cout, ks, stride, ups, gn)
SHAPES = {
    "vae512_128_128_gn": (8, 128, 0, 512, 512, 128, 3, 1, 0, 1),
    "vae512_128_128_nogn": (8, 128, 0, 512, 512, 128, 3, 1, 0, 0),
    "vae256_256_256_nogn": (8, 256, 0, 256, 256, 256, 3, 1, 0, 0),
    "vae512_256_128_gn": (8, 256, 0, 512, 512, 128, 3, 1, 0, 1),
    "vae512_up_256_256": (8, 256, 0, 256, 256, 256, 3, 1, 1, 0),
    "vae256_up_512_512": (8, 512, 0, 128, 128, 512, 3, 1, 1, 0),
    "vae128_up_512_512": (8, 512, 0, 64, 64, 512, 3, 1, 1, 0),
    "vae256_256_256_gn": (8, 256, 0, 256, 256, 256, 3, 1, 0, 1),
    "vae128_512_512_gn": (8, 512, 0, 128, 128, 512, 3, 1, 0, 1),
    "vae64_512_512_gn": (8, 512, 0, 64, 64, 512, 3, 1, 0, 1),
    "vae512_128_3_gn": (8, 128, 0, 512, 512, 3, 3, 1, 0, 1),
    "unetL0_320_320_gn": (8, 320, 0, 64, 64, 320, 3, 1, 0, 1),
    # (not UNet shapes: N % 128 == 0, for kernel-against-kernel probes of the UNet-sized maps -- LDIFF_CONV3X3_DATAFLOW=0 / 2)
    "probe_L0_320_256_gn": (8, 320, 0, 64, 64, 256, 3, 1, 0, 1),
    "probe_L0_640_256_gn": (8, 640, 0, 64, 64, 256, 3, 1, 0, 1),
    "probe_L1_640_640_gn": (8, 640, 0, 32, 32, 640, 3, 1, 0, 1),
    "unetL0_cat640_320_gn": (8, 320, 320, 64, 64, 320, 3, 1, 0, 1),
    "unetL1_640_640_gn": (8, 640, 0, 32, 32, 640, 3, 1, 0, 1),
    "unetL1_cat1280_640_gn": (8, 640, 640, 32, 32, 640, 3, 1, 0, 1),
    "unetL2_1280_1280_gn": (8, 1280, 0, 16, 16, 1280, 3, 1, 0, 1),
    "unetL2_cat2560_1280_gn": (8, 1280, 1280, 16, 16, 1280, 3, 1, 0, 1),
    "unetL3_1280_1280_gn": (8, 1280, 0, 8, 8, 1280, 3, 1, 0, 1),
    "unetL3_cat2560_1280_gn": (8, 1280, 1280, 8, 8, 1280, 3, 1, 0, 1),
    "lin_L0_qkv_320_960": (1, 320, 0, 1, 32768, 960, 1, 1, 0, 0),
    "lin_L0_ff1_320_2560": (1, 320, 0, 1, 32768, 2560, 1, 1, 0, 0),
    "lin_L0_ff2_1280_320": (1, 1280, 0, 1, 32768, 320, 1, 1, 0, 0),
    "lin_L1_ff1_640_5120": (1, 640, 0, 1, 8192, 5120, 1, 1, 0, 0),
    "lin_L1_qkv_640_1920": (1, 640, 0, 1, 8192, 1920, 1, 1, 0, 0),
    "lin_L1_out_640_640": (1, 640, 0, 1, 8192, 640, 1, 1, 0, 0),
    "lin_L1_ff2_2560_640": (1, 2560, 0, 1, 8192, 640, 1, 1, 0, 0),
    "lin_L2_qkv_1280_3840": (1, 1280, 0, 1, 2048, 3840, 1, 1, 0, 0),
    "lin_L2_out_1280_1280": (1, 1280, 0, 1, 2048, 1280, 1, 1, 0, 0),
    "lin_L0_out_320_320": (1, 320, 0, 1, 32768, 320, 1, 1, 0, 0),
    "lin_L2_ff1_1280_10240": (1, 1280, 0, 1, 2048, 10240, 1, 1, 0, 0),
    "lin_L2_ff2_5120_1280": (1, 5120, 0, 1, 2048, 1280, 1, 1, 0, 0),
    "lin_L3_qkv_1280_3840": (1, 1280, 0, 1, 512, 3840, 1, 1, 0, 0),
    "lin_L3_ff1_1280_10240": (1, 1280, 0, 1, 512, 10240, 1, 1, 0, 0),
    "lin_L3_ff2_5120_1280": (1, 5120, 0, 1, 512, 1280, 1, 1, 0, 0),
    "lin_L1_ff1_640_5120b": (1, 640, 0, 1, 8192, 5120, 1, 1, 0, 0),
    "conv1x1_L0_320_320_gn": (8, 320, 0, 64, 64, 320, 1, 1, 0, 1),
    "down_L0_320_320_s2": (8, 320, 0, 64, 64, 320, 3, 2, 0, 0),
    "down_L1_640_640_s2": (8, 640, 0, 32, 32, 640, 3, 2, 0, 0),
    "down_L2_1280_1280_s2": (8, 1280, 0, 16, 16, 1280, 3, 2, 0, 0),
    "down_L2x2_2560_1280_s2": (8, 2560, 0, 16, 16, 1280, 3, 2, 0, 0),
    "down_L1x2_1280_640_s2": (8, 1280, 0, 32, 32, 640, 3, 2, 0, 0),
    "down_L0x2_640_320_s2": (8, 640, 0, 64, 64, 320, 3, 2, 0, 0),
    "down_vae_128_128_s2": (8, 128, 0, 512, 512, 128, 3, 2, 0, 0),
    "down_vae_256_256_s2": (8, 256, 0, 256, 256, 256, 3, 2, 0, 0),
    "down_vae_512_512_s2": (8, 512, 0, 128, 128, 512, 3, 2, 0, 0),
}
# name: (B, heads, Lq, Lk, d)
ATTN = {
    "attn_L0_self_d40": (8, 8, 4096, 4096, 40),
    "attn_L1_self_d80": (8, 8, 1024, 1024, 80),
    "attn_L2_self_d160": (8, 8, 256, 256, 160),
    "attn_L0_cross_d40_L6": (8, 8, 4096, 6, 40),
    "attn_vae_d512": (8, 1, 4096, 4096, 512),
}


def time_it(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    if os.environ.get("LDIFF_LIB"):   # diagnostic builds (ablations): never the product library
        _lib.LIB_PATH = os.path.abspath(os.environ["LDIFF_LIB"])
    lib = _lib.load()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, (B, C1, C2, H, W, Cout, ks, stride, ups, gn) in SHAPES.items():
        if flt and not any(f in name for f in flt):
            continue
        Cin = C1 + C2
        He, We = H << ups, W << ups
        Ho, Wo = (He + 2 * (ks // 2) - ks) // stride + 1, (We + 2 * (ks // 2) - ks) // stride + 1
        x = torch.randn((B, H, W, C1), device=DEV, dtype=torch.float16)
        x2 = torch.randn((B, H, W, C2), device=DEV, dtype=torch.float16) if C2 else None
        Nrows, Nst = (Cout + 15) // 16 * 16, (Cout + 3) // 4 * 4
        w = (torch.randn((Nrows, ks * ks * Cin), device=DEV) / math.sqrt(ks * ks * Cin)).to(torch.float16)
        y = torch.empty((B, Ho, Wo, Nst), device=DEV, dtype=torch.float16)
        bias = torch.randn(Nrows, device=DEV)
        sc, sh = torch.rand((B, Cin), device=DEV) + 0.5, torch.randn((B, Cin), device=DEV) * 0.1
        a = _lib.ConvArgs()
        a.x, a.C1 = x.data_ptr(), C1
        if C2:
            a.x2, a.C2 = x2.data_ptr(), C2
        a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride, a.pad_t, a.pad_l, a.ups = B, H, W, Ho, Wo, ks, stride, ks // 2, ks // 2, ups
        a.w, a.N, a.Nrows, a.bias = w.data_ptr(), Nst, Nrows, bias.data_ptr()
        a.n_real = Cout
        if gn:
            a.gn_scale, a.gn_shift, a.silu_in = sc.data_ptr(), sh.data_ptr(), int(os.environ.get("LDIFF_BENCH_SILU", "1"))
        a.y, a.ldy = y.data_ptr(), Nst
        if os.environ.get('LDIFF_BENCH_RES'):   # residual operand (conv2 of a resnet block)
            res = torch.randn((B, Ho, Wo, Nst), device=DEV, dtype=torch.float16)
            a.res, a.ld_res = res.data_ptr(), Nst
        if os.environ.get('LDIFF_BENCH_STATS'):
            R = lib.ldiff_op_conv_stats_blocks(C.byref(a))
            if R > 0:
                st = torch.empty((B, Nst, R, 2), device=DEV)
                a.stats = st.data_ptr()
        ms = time_it(lambda: _lib.check(lib.ldiff_op_conv(C.byref(a), sp)), iters)
        fl = 2.0 * B * Ho * Wo * Cout * ks * ks * Cin
        print(f"{name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  ({fl/1e9:.1f} GFLOP)", flush=True)
    for name, (B, heads, Lq, Lk, d) in ATTN.items():
        if flt and not any(f in name for f in flt):
            continue
        Cc = heads * d
        q = torch.randn((B, Lq, Cc), device=DEV, dtype=torch.float16)
        k = torch.randn((B, Lk, Cc), device=DEV, dtype=torch.float16)
        v = torch.randn((B, Lk, Cc), device=DEV, dtype=torch.float16)
        o = torch.empty_like(q)
        fn = lambda: _lib.check(lib.ldiff_op_attention(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, o.data_ptr(), Cc, B, heads, Lq, Lk,
                                                       d, Lq * Cc, Lk * Cc, Lq * Cc, 1 / math.sqrt(d), sp))
        ms = time_it(fn, iters)
        fl = 4.0 * B * heads * Lq * Lk * d
        print(f"{name:28s} {ms*1e3:9.1f} us  {fl/ms/1e9:8.1f} TFLOP/s  ({fl/1e9:.1f} GFLOP)", flush=True)


if __name__ == "__main__":
    main()
