import ctypes as C, math, sys, os
sys.path.insert(0, "/root/repo")
import torch
from ldiffusion_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, Cc = 32768, 320
g = torch.Generator().manual_seed(0)
xs = (torch.randn((M, 2 * Cc), generator=g)).to(torch.float16).to(DEV)
gamma, beta = torch.ones(Cc).to(DEV), torch.zeros(Cc).to(DEV)
for N, geglu in ((960, 0), (320, 0), (2560, 1)):
    w = (torch.randn((N, Cc), generator=g) / math.sqrt(Cc)).to(torch.float16).to(DEV)
    b = torch.zeros(N).to(DEV)
    y = torch.empty((M, N // 2 if geglu else N), dtype=torch.float16, device=DEV)
    for _ in range(3):
        _lib.check(lib.ldiff_op_ln_linear(xs.data_ptr(), 2 * Cc, Cc, M, Cc, gamma.data_ptr(), beta.data_ptr(), 1e-5, w.data_ptr(), N, N, b.data_ptr(), geglu, y.data_ptr(), y.shape[1], 0, 1.0, sp()))
    torch.cuda.synchronize()
