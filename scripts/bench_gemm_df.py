"""A/B timing of the UNet's 1x1 / linear launches (B = 8, 512x512: M = 32768 / 8192 / 2048 rows at C = 320 / 640 / 1280) with their real epilogues:
the LDS-DMA GEMM (ldiff_conv_args.gemm_df = -1) against the producer / consumer GEMM (kernels_gemm_df.hip) at the launcher's plan (gemm_df = 1)
and, with --plans, at every unit shape.  Interleaved rounds in ONE process (cdna guide rule 24), HIP events on the launch stream, random data;
the fragment packing is cached (LDIFF_OP_CACHE_FRAG=1), as the executors cache it per layer.
usage: python scripts/bench_gemm_df.py [filter-substring ...] [--plans] [--rounds N] [--iters N]
"""
import ctypes as C
import math
import os
import sys

os.environ.setdefault("LDIFF_OP_CACHE_FRAG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ldiffusion_amd import _lib

DEV = "cuda:0"
# name: (M, K, N, epilogue)   epilogue: plain | plain_nobias | split_out | split_res_out | geglu
SHAPES = {
    "L0_out_320_320": (32768, 320, 320, "split_res_out"),
    "L0_ff2_1280_320": (32768, 1280, 320, "split_res_out"),
    "L0_projin_2x320_320": (32768, 640, 320, "split_out"),
    "L0_projout_2x320_320": (32768, 640, 320, "split_res_out"),
    "L0_qkv_320_960": (32768, 320, 960, "plain_nobias"),
    "L0_q2_320_320": (32768, 320, 320, "plain_nobias"),
    "L0_ff1_320_2560": (32768, 320, 2560, "geglu"),
    "L0_sc_2x960_320": (32768, 1920, 320, "split_out"),
    "L0_sc_2x640_320": (32768, 1280, 320, "split_out"),
    "L1_out_640_640": (8192, 640, 640, "split_res_out"),
    "L1_ff2_2560_640": (8192, 2560, 640, "split_res_out"),
    "L1_proj_2x640_640": (8192, 1280, 640, "split_out"),
    "L1_qkv_640_1920": (8192, 640, 1920, "plain_nobias"),
    "L1_q2_640_640": (8192, 640, 640, "plain_nobias"),
    "L1_ff1_640_5120": (8192, 640, 5120, "geglu"),
    "L1_sc_2x1920_640": (8192, 3840, 640, "split_out"),
    "L2_out_1280_1280": (2048, 1280, 1280, "split_res_out"),
    "L2_ff2_5120_1280": (2048, 5120, 1280, "split_res_out"),
    "L2_proj_2x1280_1280": (2048, 2560, 1280, "split_out"),
    "L2_qkv_1280_3840": (2048, 1280, 3840, "plain_nobias"),
    "L2_q2_1280_1280": (2048, 1280, 1280, "plain_nobias"),
    "L2_ff1_1280_10240": (2048, 1280, 10240, "geglu"),
}
PLANS = {"8x5": 16 * 8 + 5, "8x4": 16 * 8 + 4, "8x2": 16 * 8 + 2, "4x5": 16 * 4 + 5, "4x4": 16 * 4 + 4, "4x2": 16 * 4 + 2}


def main():
    flt = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 5
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    all_plans = "--plans" in sys.argv
    if os.environ.get("LDIFF_LIB"):   # diagnostic builds: never the product library
        _lib.LIB_PATH = os.path.abspath(os.environ["LDIFF_LIB"])
    lib = _lib.load()
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot = {}
    for name, (M, K, N, epi) in SHAPES.items():
        if flt and not any(f in name for f in flt):
            continue
        geglu = epi == "geglu"
        x = torch.randn((M, K), device=DEV, dtype=torch.float16)
        w = (torch.randn((N, K), device=DEV) / math.sqrt(K)).to(torch.float16)
        bias = torch.randn(N, device=DEV)
        Nout = N // 2 if geglu else N
        split = epi in ("split_out", "split_res_out")
        y = torch.empty((M, 2 * Nout if split else Nout), device=DEV, dtype=torch.float16)
        a = _lib.ConvArgs()
        a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout, a.ks, a.stride = x.data_ptr(), K, 1, 1, M, 1, M, 1, 1
        a.w, a.N, a.Nrows = w.data_ptr(), N, N
        if epi != "plain_nobias":
            a.bias = bias.data_ptr()
        a.y, a.ldy, a.y_lo, a.geglu = y.data_ptr(), y.shape[1], Nout if split else 0, int(geglu)
        if epi == "split_res_out":
            res = torch.randn((M, 2 * N), device=DEV, dtype=torch.float16)
            a.res, a.ld_res, a.res_lo = res.data_ptr(), 2 * N, N
        variants = {"dma": -1, "df": 1}
        if all_plans:
            variants.update(PLANS)
        best = {k: 1e9 for k in variants}
        med = {k: [] for k in variants}
        for _ in range(rounds):
            for k, v in variants.items():
                a.gemm_df = v
                _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _i in range(iters):
                    lib.ldiff_op_conv(C.byref(a), sp)
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / iters * 1e3
                best[k] = min(best[k], t); med[k].append(t)
        fl = 2.0 * M * N * K
        row = "  ".join(f"{k} {sorted(med[k])[len(med[k]) // 2]:6.1f} ({fl / sorted(med[k])[len(med[k]) // 2] / 1e6:5.0f} TF)" for k in variants)
        print(f"{name:22s} {fl / 1e9:6.1f} GF  {row}", flush=True)
        for k in variants:
            tot[k] = tot.get(k, 0.0) + sorted(med[k])[len(med[k]) // 2]
    print("sum of medians (us): " + "  ".join(f"{k} {v:.1f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
