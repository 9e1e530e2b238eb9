"""In-kernel stamps of the dataflow GEMM (diagnostic build -DGDF_STAMPS; never the product library): where one consumer, one loader and one
epilogue wave of workgroup 0 spend their cycles.  Build + run:
  python scripts/gemm_df_stamps.py --build          (here or on the GPU box: hipcc cross-compiles)
  python scripts/gemm_df_stamps.py [shape-filter]   (on the GPU box)
"""
import ctypes as C, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "stamps")
DEFS = [a[6:] for a in sys.argv[1:] if a.startswith("--def=")]       # e.g. --def=GDF_TUNE --def=GDF_RING=3 --def=GDF_SLICES=3
ABL = [a[6:] for a in sys.argv[1:] if a.startswith("--abl=")]       # e.g. --abl=NOW --abl=NOX: ablation builds (timing only, results wrong)
TAG = "".join("_" + a.lower() for a in ABL) + "".join("_" + d.lower().replace("gdf_", "").replace("=", "") for d in DEFS)
LIB = os.path.join(OUT, f"libldiff_hip_gdf_stamps{TAG}.so")
if "--build" in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, f"kernels_gemm_df{TAG}.o")
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-comment", "-DGDF_STAMPS"] + [f"-DGDF_ABL_{a}" for a in ABL] + [f"-D{d}" for d in DEFS] + ["-c",
                    os.path.join(ROOT, "ldiffusion_amd", "csrc", "kernels_gemm_df.hip"), "-o", obj], check=True)
    others = [os.path.join(ROOT, "build", "obj", f) for f in os.listdir(os.path.join(ROOT, "build", "obj")) if f.endswith(".o") and f != "kernels_gemm_df.o"]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + others, check=True)
    print("built", LIB)
    sys.exit(0)
sys.path.insert(0, ROOT)
os.environ["LDIFF_OP_CACHE_FRAG"] = "1"
import torch
from ldiffusion_amd import _lib
_lib.LIB_PATH = LIB
lib = _lib.load()
raw = C.CDLL(LIB)
DEV = "cuda:0"
sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = {"L0_out": (32768, 320, 320, "split_res_out", 133), "L0_qkv": (32768, 320, 960, "plain_nobias", 133), "L0_ff1": (32768, 320, 2560, "geglu", 133),
          "L0_ff2": (32768, 1280, 320, "split_res_out", 133), "L1_ff1": (8192, 640, 5120, "geglu", 132), "L1_qkv": (8192, 640, 1920, "plain_nobias", 132),
          "L2_ff1": (2048, 1280, 10240, "geglu", 133), "L1_out_4x5": (8192, 640, 640, "split_res_out", 69)}
flt = [a for a in sys.argv[1:] if not a.startswith("--")]
for name, (M, K, N, epi, plan) in SHAPES.items():
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
    a.y, a.ldy, a.y_lo, a.geglu, a.gemm_df = y.data_ptr(), y.shape[1], Nout if split else 0, int(geglu), plan
    if epi == "split_res_out":
        res = torch.randn((M, 2 * N), device=DEV, dtype=torch.float16)
        a.res, a.ld_res, a.res_lo = res.data_ptr(), 2 * N, N
    for _ in range(3):
        _lib.check(lib.ldiff_op_conv(C.byref(a), sp))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.ldiff_op_conv(C.byref(a), sp)
    e1.record()
    torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 100.0
    buf = (C.c_ulonglong * 64)()
    assert raw.ldiff_debug_gdf_stamps(buf) == 0
    c, l, e = list(buf[0:16]), list(buf[16:32]), list(buf[32:48])
    mt, ntw = plan >> 4, plan & 15
    nk = K // 64
    units = max(c[4], 1)
    mfma_cycles = units * nk * mt * ntw * 2 * 16
    print(f"[{TAG or 'full'}] {name} M={M} K={K} N={N} {epi} plan {mt}x{ntw}: workgroup 0 ran {c[4]} units ({nk} K steps each); {wall_us:.1f} us per launch")
    print(f"  consumer: total {c[6]}  = prologue {c[0]} + K loops {c[6] - c[0] - c[3] - c[5]} (matrix cycles {mfma_cycles}) + staging {c[3]} + next-unit wait {c[5]};  "
          f"polls: steps {c[1]}, slices {c[2]}")
    print(f"  loader 0: total {l[0]}  vmcnt waits {l[1]}  consumer waits {l[2]} ({l[4]} polls)  issue {l[3]}  steps {l[5]}")
    print(f"  epilogue 0: total {e[0]}  slice waits {e[1]} ({e[3]} polls)  read + arithmetic + stores {e[2]}  slices {e[4]}")
