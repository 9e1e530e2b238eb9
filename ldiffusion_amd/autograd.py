"""torch.autograd.Function wrappers over the forward / backward HIP kernels: the tape is torch's, every contraction, normalisation,
activation and attention FLOP of both passes runs in libldiff_hip.so (include/ldiff.h, "Backward-pass primitives").

This is the arithmetic layer of the reference's fine-tuning step (/root/reference/ldiffusion.py:227-255: V5 loop, `engine.backward`,
`engine.step`): UNet2DConditionModel + the 768->768 text projection are trained on 64 x 64 images = 8 x 8 latents, where a step is bound
by the weights it reads and writes.  Conventions: activations and their gradients are NHWC float16 CUDA tensors, parameters are float32
masters in the torch / diffusers layouts ([Cout, Cin, k, k], [out, in]) and receive float32 gradients.
  * dgrad = the forward conv kernel on the weights rearranged to [Cin][k][k][Cout] with the taps flipped (a layout cast in torch),
  * wgrad = the forward GEMM kernel over K = M on dy^T (ldiff_op_transpose) and im2col(x)^T (ldiff_op_im2col_t),
  * GroupNorm(+SiLU), LayerNorm, GEGLU, attention: dedicated backward kernels (csrc/kernels_bwd.hip).
There is no CPU or torch-op fallback for those; reshapes / layout casts / the residual `+` are torch tensor plumbing.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _r(x, m):
    return (x + m - 1) // m * m


def _check_act(x, what):
    if not (x.is_cuda and x.dtype == torch.float16 and x.is_contiguous()):
        raise ValueError(f"{what} must be a contiguous float16 CUDA tensor")


_PACK_CACHE: dict | None = None   # (weight data_ptr, dgrad) -> packed tensor, while a PackPlan is active (train.GraphedStep)


def _pack_dims(Cout, Cin, dgrad, cols=None):
    if dgrad:
        return _r(_r(Cin, 8), 16), (cols if cols is not None else _r(Cout, 8))
    return _r(Cout, 16), _r(Cin, 8)


def pack_weight(w: torch.Tensor, dgrad: bool = False, cols: int | None = None) -> torch.Tensor:
    """[Cout, Cin, k, k] (or [out, in]) float32 -> the kernel layout, float16, one launch (ldiff_op_pack_weight):
    forward  [roundup(Cout,16)][k*k*roundup(Cin,8)]        rows >= Cout zero;
    dgrad    [roundup(cols_x,16)][k*k*cols]  = the weights rearranged to [Cin][k][k][Cout] with the taps flipped (`cols` = channels of dy).
    While a PackPlan is active, the layouts it has already produced for this storage are returned instead."""
    lib = _lib.load()
    w = w.detach()
    if w.dim() == 2:
        w = w[:, :, None, None]
    if _PACK_CACHE is not None:
        hit = _PACK_CACHE.get((w.data_ptr(), bool(dgrad)))
        if hit is not None and (not dgrad or cols is None or hit.shape[1] == w.shape[2] * w.shape[3] * cols):
            return hit
    if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
        w = w.float().contiguous()
    Cout, Cin, k, _ = w.shape
    rows, cp = _pack_dims(Cout, Cin, dgrad, cols)
    out = torch.empty((rows, k * k * cp), dtype=torch.float16, device=w.device)
    _lib.check(lib.ldiff_op_pack_weight(w.data_ptr(), out.data_ptr(), Cout, Cin, k, rows, cp, int(dgrad), _sp()))
    return out


class PackPlan:
    """Forward and dgrad layouts of MANY conv / linear weights ([Cout, Cin, k, k] or [out, in], contiguous float32 CUDA) in ONE launch
    (ldiff_op_pack_weight_multi) into persistent float16 buffers; `run()` refreshes them from the current master values, `cache()` is the
    (data_ptr, dgrad) -> packed tensor map `pack_weight` consults while the plan is active (`with plan:`)."""

    def __init__(self, weights):
        import numpy as np
        ws = [w for w in weights if w.dim() in (2, 4)]
        if not ws:
            raise ValueError("PackPlan: no weights")
        dev = ws[0].device
        metas, total = [], 0
        for w in ws:
            if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()):
                raise ValueError("PackPlan: weights must be contiguous float32 CUDA tensors")
            Cout, Cin = w.shape[0], w.shape[1]
            k = w.shape[2] if w.dim() == 4 else 1
            if k not in (1, 3):
                raise ValueError("PackPlan: kernel size 1 or 3")
            for dgrad in (False, True):
                rows, cp = _pack_dims(Cout, Cin, dgrad)
                metas.append((w, dgrad, Cout, Cin, k * k, rows, cp, total))
                total += rows * k * k * cp
        self.flat = torch.empty(total, dtype=torch.float16, device=dev)
        self._cache, rec, prefix, tiles = {}, [], [0], 0
        for w, dgrad, Cout, Cin, kk, rows, cp, off in metas:
            view = self.flat[off:off + rows * kk * cp].view(rows, kk * cp)
            self._cache[(w.data_ptr(), dgrad)] = view
            if dgrad:
                tx, ty = -(-cp // 64), -(-rows // (64 if kk == 1 else 16))
            else:
                tx, ty = -(-cp // 256), (-(-rows // 8) if kk == 1 else rows)
            rec.append((w.data_ptr(), view.data_ptr(), Cout, Cin, kk, rows, cp, int(dgrad), tx, 0))
            tiles += tx * ty
            prefix.append(tiles)
        arr = np.zeros(len(rec), dtype=np.dtype([("w", "<u8"), ("dst", "<u8"), ("i", "<i4", (8,))]))
        for j, r in enumerate(rec):
            arr[j] = (r[0], r[1], r[2:])
        self.entries = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(dev)
        self.prefix = torch.tensor(prefix, dtype=torch.int32).to(dev)
        self.n_entries, self.n_tiles = len(rec), tiles
        self._keep = ws

    def run(self):
        _lib.check(_lib.load().ldiff_op_pack_weight_multi(self.entries.data_ptr(), self.prefix.data_ptr(), self.n_entries, self.n_tiles, _sp()))

    def cache(self):
        return self._cache


class packed_weights:
    """`with packed_weights(plan_a, plan_b, ...):` -- pack_weight returns the plans' layouts for the storages they cover."""

    def __init__(self, *plans):
        self.map = {}
        for p in plans:
            self.map.update(p.cache())

    def __enter__(self):
        global _PACK_CACHE
        self.prev, _PACK_CACHE = _PACK_CACHE, self.map
        return self

    def __exit__(self, *exc):
        global _PACK_CACHE
        _PACK_CACHE = self.prev
        return False


def _conv_call(x, w16, Cout, k, stride, ups, bias=None, Ho=None, Wo=None, out_f32=False, pad=None):
    """ldiff_op_conv on NHWC x [B,H,W,C] with packed weights; returns [B,Ho,Wo,roundup(Cout,8)] f16 (or f32 [.., roundup(Cout,4)])."""
    lib = _lib.load()
    B, H, W, Cin = x.shape
    pad = k // 2 if pad is None else pad
    He, We = H << ups, W << ups
    Ho = (He + 2 * pad - k) // stride + 1 if Ho is None else Ho
    Wo = (We + 2 * pad - k) // stride + 1 if Wo is None else Wo
    a = _lib.ConvArgs()
    a.x, a.C1, a.B, a.Hin, a.Win, a.Hout, a.Wout = x.data_ptr(), Cin, B, H, W, Ho, Wo
    a.ks, a.stride, a.pad_t, a.pad_l, a.ups = k, stride, pad, pad, ups
    a.w, a.N, a.Nrows = w16.data_ptr(), _r(Cout, 4), w16.shape[0]
    keep = []
    if bias is not None:
        b = bias.detach()
        if not (b.dtype == torch.float32 and b.is_contiguous() and b.numel() == w16.shape[0]):   # the kernel reads Nrows floats
            bp = torch.zeros(w16.shape[0], dtype=torch.float32, device=x.device)
            bp[:Cout] = b.float()
            b = bp
        a.bias = b.data_ptr()
        keep.append(b)
    ld = _r(Cout, 4) if out_f32 else _r(Cout, 8)
    alloc = torch.empty if ld == _r(Cout, 4) else torch.zeros   # the kernel writes roundup(Cout, 4) columns; pad columns beyond must be zero
    y = alloc((B, Ho, Wo, ld), dtype=torch.float32 if out_f32 else torch.float16, device=x.device)
    a.y, a.ldy, a.out_f32 = y.data_ptr(), ld, int(out_f32)
    _lib.check(lib.ldiff_op_conv(C.byref(a), _sp()))
    return y


class Conv2dFn(torch.autograd.Function):
    """y = conv2d(nearest_up(x, 2**ups), weight, bias, stride, padding=k//2) on NHWC float16 activations (k in {1, 3}).
    Channel counts of x must be multiples of 8 (pad the 4-channel latents); y has roundup(Cout, 8) channels, the pad columns zero."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride=1, ups=0):
        _check_act(x, "x")
        Cout, Cin, k = weight.shape[0], weight.shape[1], (weight.shape[2] if weight.dim() == 4 else 1)
        if x.shape[-1] != _r(Cin, 8):
            raise ValueError(f"x has {x.shape[-1]} channels, the weight expects {Cin} (padded to {_r(Cin, 8)})")
        y = _conv_call(x, pack_weight(weight), Cout, k, stride, ups, bias)
        ctx.save_for_backward(x, weight)
        ctx.meta = (stride, ups, bias is not None, k, Cout, Cin)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weight = ctx.saved_tensors
        stride, ups, has_bias, k, Cout, Cin = ctx.meta
        dy = dy.contiguous()
        B, H, W, Cx = x.shape
        _, Ho, Wo, Cy = dy.shape
        w4 = weight if weight.dim() == 4 else weight[:, :, None, None]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dgrad: conv (stride 1) of dy -- zero-inserted for a stride-2 forward -- with the weights [Cin][k][k][Cout], taps flipped
            wd = pack_weight(w4, dgrad=True, cols=Cy)
            g = dy
            He, We = H << ups, W << ups
            if stride == 2:
                g = torch.zeros((B, He, We, Cy), dtype=torch.float16, device=x.device)
                g[:, ::2, ::2] = dy
            dxu = _conv_call(g, wd, Cx, k, 1, 0)
            dx = dxu if ups == 0 else dxu.view(B, H, 2, W, 2, Cx).float().sum((2, 4)).to(torch.float16)
        if ctx.needs_input_grad[1]:
            # wgrad: dW[n][tap*Cx + c] = sum_m dy[m, n] * xcol[m, tap*Cx + c]  as a GEMM over K = M
            M = B * Ho * Wo
            Mpad = _r(M, 8)
            Kc = k * k * Cx
            dyT = torch.empty((Cy, Mpad), dtype=torch.float16, device=x.device)
            _lib.check(lib.ldiff_op_transpose(dy.data_ptr(), dyT.data_ptr(), M, Cy, Cy, Mpad, _sp()))
            xcolT = torch.empty((_r(Kc, 16), Mpad), dtype=torch.float16, device=x.device)   # im2col_t writes rows < Kc (columns >= M zero)
            if xcolT.shape[0] > Kc:
                xcolT[Kc:].zero_()
            _lib.check(lib.ldiff_op_im2col_t(x.data_ptr(), xcolT.data_ptr(), B, H, W, Cx, k, stride, k // 2, ups, Ho, Wo, Mpad, _sp()))
            g = _conv_call(dyT.view(1, 1, Cy, Mpad), xcolT, Kc, 1, 1, 0, out_f32=True)        # [1,1,Cy,Kc] f32
            if k == 1 and Cx == Cin and Cy == Cout and g.shape[-1] == Cin:
                dw = g.view(weight.shape)   # 1 x 1 / linear with unpadded channel counts: the GEMM's [Cout][Cin] output IS the weight gradient (no unpack launch)
            else:
                dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                _lib.check(lib.ldiff_op_unpack_wgrad(g.data_ptr(), dw.data_ptr(), Cout, Cin, k, Cx, g.shape[-1], _sp()))
        if has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(Cy, dtype=torch.float32, device=x.device)
            _lib.check(lib.ldiff_op_colsum(dy.data_ptr(), db.data_ptr(), B * Ho * Wo, Cy, Cy, _sp()))
            db = db[:Cout]
        return dx, dw, db, None, None


def linear(x, weight, bias=None):
    """F.linear on float16 rows [..., Cin] through the conv kernels (k = 1)."""
    shp = x.shape
    y = Conv2dFn.apply(x.reshape(1, 1, -1, shp[-1]).contiguous(), weight, bias, 1, 0)
    y = y.reshape(*shp[:-1], y.shape[-1])
    return y if y.shape[-1] == weight.shape[0] else y[..., :weight.shape[0]]


class GroupNormFn(torch.autograd.Function):
    """y = act(group_norm(x, groups, gamma, beta, eps)) on NHWC float16 [B, H, W, C]; act = SiLU when `silu`."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu):
        _check_act(x, "x")
        lib = _lib.load()
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((B, groups), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _lib.check(lib.ldiff_op_gn_train_fwd(x.data_ptr(), y.data_ptr(), g32.data_ptr(), b32.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, H * W, Cc,
                                             groups, float(eps), int(silu), _sp()))
        ctx.save_for_backward(x, g32, b32, mean, rstd)
        ctx.meta = (groups, int(silu))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, g32, b32, mean, rstd = ctx.saved_tensors
        groups, silu = ctx.meta
        B, H, W, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg, db = torch.zeros(Cc, dtype=torch.float32, device=x.device), torch.zeros(Cc, dtype=torch.float32, device=x.device)
        _lib.check(lib.ldiff_op_gn_train_bwd(x.data_ptr(), dy.data_ptr(), g32.data_ptr(), b32.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                             dg.data_ptr(), db.data_ptr(), B, H * W, Cc, groups, silu, _sp()))
        return dx, dg, db, None, None, None


class LayerNormFn(torch.autograd.Function):
    """F.layer_norm over the last dim of float16 rows [..., C]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _check_act(x, "x")
        lib = _lib.load()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        y = torch.empty_like(x)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _lib.check(lib.ldiff_op_layernorm(x.data_ptr(), 0, 0, y.data_ptr(), rows, Cc, g32.data_ptr(), b32.data_ptr(), float(eps), _sp()))
        ctx.save_for_backward(x, g32)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, g32 = ctx.saved_tensors
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg, db = torch.zeros(Cc, dtype=torch.float32, device=x.device), torch.zeros(Cc, dtype=torch.float32, device=x.device)
        _lib.check(lib.ldiff_op_ln_bwd(x.data_ptr(), dy.data_ptr(), g32.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, Cc, ctx.eps, _sp()))
        return dx, dg, db, None


class GegluFn(torch.autograd.Function):
    """[M, 2*C4] = [h | gate] -> h * gelu_erf(gate)   (diffusers GEGLU)."""

    @staticmethod
    def forward(ctx, x):
        _check_act(x, "x")
        lib = _lib.load()
        C4 = x.shape[-1] // 2
        M = x.numel() // x.shape[-1]
        y = torch.empty(x.shape[:-1] + (C4,), dtype=torch.float16, device=x.device)
        _lib.check(lib.ldiff_op_geglu(x.data_ptr(), y.data_ptr(), M, C4, _sp()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        C4 = x.shape[-1] // 2
        M = x.numel() // x.shape[-1]
        dx = torch.empty_like(x)
        _lib.check(lib.ldiff_op_geglu_bwd(x.data_ptr(), dy.contiguous().data_ptr(), dx.data_ptr(), M, C4, _sp()))
        return dx


class SiluFn(torch.autograd.Function):
    """y = x * sigmoid(x) on float16 tensors of any shape (ldiff_op_silu / ldiff_op_silu_bwd): the two activations of the time-embedding MLP,
    the last elementwise ops of the step that used to run through ATen."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _check_act(x, "silu input")
        y = torch.empty_like(x)
        _lib.check(_lib.load().ldiff_op_silu(x.data_ptr(), y.data_ptr(), x.numel(), _sp()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(_lib.load().ldiff_op_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _sp()))
        return dx


def silu(x):
    return SiluFn.apply(x)


class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v per head on float16 [B, L, heads*d] tensors (K/V given per batch entry, not broadcast)."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        for t, n in ((q, "q"), (k, "k"), (v, "v")):
            _check_act(t, n)
        lib = _lib.load()
        B, Lq, Cc = q.shape
        Lk = k.shape[1]
        d = Cc // heads
        o = torch.empty_like(q)
        scale = 1.0 / d ** 0.5
        _lib.check(lib.ldiff_op_attention(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, o.data_ptr(), Cc, B, heads, Lq, Lk, d, Lq * Cc, Lk * Cc, Lq * Cc,
                                          scale, _sp()))
        ctx.save_for_backward(q, k, v)
        ctx.meta = (heads, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        lib = _lib.load()
        q, k, v = ctx.saved_tensors
        heads, scale = ctx.meta
        B, Lq, Cc = q.shape
        Lk = k.shape[1]
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        _lib.check(lib.ldiff_op_attention_bwd(q.data_ptr(), Cc, k.data_ptr(), Cc, v.data_ptr(), Cc, do.data_ptr(), Cc, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                              B, heads, Lq, Lk, Cc // heads, Lq * Cc, Lk * Cc, Lq * Cc, scale, _sp()))
        return dq, dk, dv, None


class InfoNceFn(torch.autograd.Function):
    """Contrastive feature loss for given sample triples (/root/reference/model/loss.py:89-109): forward value and d loss / d features from
    ONE launch of ldiff_op_infonce; backward only scales the stored gradient.  features f32 [B, n, H, W]; bi/ai/pi int32 [T], ni int32 [T, K]."""

    @staticmethod
    def forward(ctx, features, bi, ai, pi, ni, temperature, count=None):
        """count: optional device int32 scalar = the number of valid triples (the index tensors are then capacity-sized: shape-stable launch)."""
        if not (features.is_cuda and features.dtype == torch.float32):
            raise ValueError("InfoNceFn: features must be a float32 CUDA tensor")
        f = features.contiguous()
        B, n, H, W = f.shape
        T, K = ni.shape
        if count is None and T > 0:
            # eager use: the indices address global reads and atomic scatters; the torch gather this replaces raised on a bad index
            # (triples sampled at another resolution than the feature map).  The capacity-sized graph path validates on the host
            # before the upload (train.GraphedStep.set_batch): no device read-back inside a replayed step.
            # ONE device reduction and ONE read-back for the four bounds (was eight host syncs per step)
            px = torch.cat([ai.reshape(-1), pi.reshape(-1), ni.reshape(-1)]).long()
            lo, hi, blo, bhi = torch.stack([px.min(), px.max(), bi.min().long(), bi.max().long()]).tolist()
            if lo < 0 or hi >= H * W or blo < 0 or bhi >= B:
                raise IndexError(f"InfoNceFn: sample indices out of range for features [B={B}, {H}x{W}] (pixel {lo}..{hi}, image {blo}..{bhi})")
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        df = torch.empty_like(f)
        _lib.check(_lib.load().ldiff_op_infonce(f.data_ptr(), B, n, H * W, bi.data_ptr(), ai.data_ptr(), pi.data_ptr(), ni.data_ptr(), T,
                                                None if count is None else count.data_ptr(), K, float(temperature), loss.data_ptr(), df.data_ptr(), _sp()))
        ctx.save_for_backward(df)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return g * df, None, None, None, None, None, None


ADAMW_CHUNK = 16384   # elements per workgroup of ldiff_op_adamw_multi (csrc/common.h)


def adamw_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
    """One AdamW update (torch.optim.AdamW semantics; the reference's DeepSpeed config, ldiffusion.py:168-171) of float32 CUDA
    parameters, in place, ALL tensors in one launch (ldiff_op_adamw_multi).  `state` is a dict the caller keeps: the step count, the two
    moment buffers per parameter and the device tables of the launch (parameter / moment pointers and the chunk list are built once;
    only the gradient pointers change from step to step)."""
    lib = _lib.load()
    live = [(i, p, g) for i, (p, g) in enumerate(zip(params, grads)) if g is not None]
    if not live:
        return   # nothing to update: the bias-correction step count must not advance either
    state["step"] = state.get("step", 0) + 1
    dev = live[0][1].device
    for i, p, g in live:
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
            raise ValueError("adamw_step: parameters must be contiguous float32 CUDA tensors")
        if i not in state:
            state[i] = (torch.zeros_like(p), torch.zeros_like(p))
    key = tuple((i, p.data_ptr(), p.numel()) for i, p, _ in live)
    tab = state.get("_tables")
    if tab is None or tab[0] != key:
        tensors, chunks = [], []
        for t, (i, p, _) in enumerate(live):
            m, v = state[i]
            tensors += [p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()]
            for first in range(0, p.numel(), ADAMW_CHUNK):
                chunks += [t, first]        # {int32 tensor, int32 pad} packed into one int64 (little endian), then int64 first
        tab = (key, torch.tensor(tensors, dtype=torch.int64).to(dev), torch.tensor(chunks, dtype=torch.int64).to(dev), len(chunks) // 2)
        state["_tables"] = tab
    keep = [g.detach().to(torch.float32).contiguous() for _, _, g in live]
    gptr = torch.tensor([g.data_ptr() for g in keep], dtype=torch.int64).to(dev)
    _lib.check(lib.ldiff_op_adamw_multi(tab[1].data_ptr(), gptr.data_ptr(), tab[2].data_ptr(), tab[3], float(lr), float(betas[0]), float(betas[1]), float(eps),
                                        float(weight_decay), int(state["step"]), _sp()))
    state["_keep"] = (keep, gptr)   # the launch is asynchronous: the pointer table and converted gradients must outlive it
