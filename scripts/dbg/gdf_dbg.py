import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ldiffusion_amd import _lib
DEV="cuda:0"
lib=_lib.load()
sp=lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(M,K,N,plan):
    g=torch.Generator().manual_seed(1)
    x=torch.randn((M,K),generator=g).to(torch.float16).to(DEV)
    w=(torch.randn((N,K),generator=g)/math.sqrt(K)).to(torch.float16)
    Nrows=(N+15)//16*16
    wd=torch.zeros((Nrows,K),dtype=torch.float16); wd[:N]=w; wd=wd.to(DEV)
    y=torch.full((M,N),float("nan"),dtype=torch.float16,device=DEV)
    a=_lib.ConvArgs()
    a.x,a.C1,a.B,a.Hin,a.Win,a.Hout,a.Wout,a.ks,a.stride=x.data_ptr(),K,1,1,M,1,M,1,1
    a.w,a.N,a.Nrows,a.y,a.ldy,a.gemm_df=wd.data_ptr(),N,Nrows,y.data_ptr(),N,plan
    _lib.check(lib.ldiff_op_conv(C.byref(a),sp())); torch.cuda.synchronize()
    ref=x.float()@wd[:N].float().t()
    yc=y.float()
    nan=torch.isnan(yc)
    bad=(~nan)&((yc-ref).abs()>2e-2*ref.abs().max())
    print(f"M={M} K={K} N={N} plan={plan>>4}x{plan&15}: nan {int(nan.sum())} bad {int(bad.sum())}")
    if nan.any():
        rows=torch.nonzero(nan.any(1)).flatten(); cols=torch.nonzero(nan.any(0)).flatten()
        print("  nan rows", rows[:10].tolist(), "...", rows[-3:].tolist(), "n", len(rows), " cols", cols[:6].tolist(), "...", cols[-3:].tolist(), "n", len(cols))
    if bad.any():
        rows=torch.nonzero(bad.any(1)).flatten(); cols=torch.nonzero(bad.any(0)).flatten()
        print("  bad rows", rows[:10].tolist(), "...", rows[-3:].tolist(), "n", len(rows), " cols", cols[:6].tolist(), "...", cols[-3:].tolist(), "n", len(cols))
for plan in (133,132,130,69,68,66):
    run(1024,320,320,plan)
    run(1024,320,384,plan)
    run(1000,128,256,plan)
