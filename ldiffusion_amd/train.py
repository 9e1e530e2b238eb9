"""Fine-tuning step of the reference on the HIP kernels (forward AND backward): trainable UNet2DConditionModel + text projection,
frozen AutoencoderKL decoder in the gradient path, the V5 feature loop and the contrastive (InfoNCE) feature loss, AdamW.

Mirrors /root/reference/ldiffusion.py:121-295 (`train_ldiffusion`): per batch
    z0 = vae.encode(Resize(64)(image)).mean                       # 8 x 8 latents                     :212,228
    for t in scheduler.timesteps:                                 # V5 loop                           :231-247
        noisy = z0 + Laplace(0, sqrt(1 - abar_t))                                                     :234-237
        den   = unet(noisy, t, proj(text_embeddings)).sample      # trainable                         :238
        rgb   = bilinear64(vae.decode(den).sample)                # frozen weights, gradient flows    :240
        gray  = (rgb * [0.2989, 0.5870, 0.1140]).sum(1)           # one feature plane per step        :241-247
    loss = InfoNceLoss.compute_loss(image, rgb_1024, features, label)                                 :251-252, model/loss.py:44-126
    engine.backward(loss); engine.step()                          # ZeRO-3 AdamW, lr 1e-5             :165-193,254-255
Graph structure follows the same diffusers restatement as csrc/model.hip / oracle/unet.py.

Scope of this module: every contraction, normalisation, activation and attention of the forward and backward pass
runs in libldiff_hip.so through ldiffusion_amd.autograd; torch provides the tape, the residual `+`, concatenation and the loss reduction
over the sampled pixel pairs.  Not built: the VGG19 content term of the loss (model/loss.py:21-42 needs ImageNet weights that are not
available offline: the term runs when a feature extractor is injected), DeepSpeed's CPU offload, bf16.  The gradient exchange is either one
flattened all-reduce over replicated parameters (`allreduce_gradients`) or reduce-scatter + all-gather around an AdamW whose state is sharded
by rank (`ShardedAdamW`).  `LDiffusionModel.train(component="ldiffusion", train_loader=...)` drives this step (ldiffusion_amd/ldiffusion.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import autograd as ag

LUMA = (0.2989, 0.5870, 0.1140)  # ldiffusion.py:241


def _nhwc16(x_nchw_f32, pad_to=8):
    B, Cc, H, W = x_nchw_f32.shape
    Cp = (Cc + pad_to - 1) // pad_to * pad_to
    out = torch.zeros((B, H, W, Cp), dtype=torch.float16, device=x_nchw_f32.device)
    out[..., :Cc] = x_nchw_f32.permute(0, 2, 3, 1)
    return out


class _Graph:
    """Parameter dict + the block builders shared by the UNet and the VAE decoder (NHWC float16 activations)."""

    def __init__(self, state_dict, device, trainable):
        self.device = torch.device(device)
        self.p = {k: torch.nn.Parameter(v.detach().to(self.device, torch.float32).contiguous(), requires_grad=trainable) for k, v in state_dict.items()}

    def parameters(self):
        return [v for v in self.p.values() if v.requires_grad]

    def named_parameters(self):
        return [(k, v) for k, v in self.p.items() if v.requires_grad]

    def conv(self, name, x, stride=1, ups=0):
        w = self.p[name + ".weight"]
        return ag.Conv2dFn.apply(x, w, self.p.get(name + ".bias"), stride, ups)   # roundup(Cout, 8) channels; pad columns (if any) are zero

    def lin(self, name, x):
        return ag.linear(x, self.p[name + ".weight"], self.p.get(name + ".bias"))

    def gn(self, name, x, groups, eps, silu):
        return ag.GroupNormFn.apply(x, self.p[name + ".weight"], self.p[name + ".bias"], groups, eps, silu)

    def ln(self, name, x):
        return ag.LayerNormFn.apply(x, self.p[name + ".weight"], self.p[name + ".bias"], 1e-5)

    def resnet(self, name, x, temb_act, groups, eps):
        """ResnetBlock2D; temb_act = SiLU(time embedding) rows [B, D] or None (VAE)."""
        h = self.gn(name + ".norm1", x, groups, eps, True)
        h = self.conv(name + ".conv1", h)
        if temb_act is not None:
            h = h + self.lin(name + ".time_emb_proj", temb_act)[:, None, None, :]
        h = self.gn(name + ".norm2", h, groups, eps, True)
        h = self.conv(name + ".conv2", h)
        if (name + ".conv_shortcut.weight") in self.p:
            x = self.conv(name + ".conv_shortcut", x)
        return x + h

    def attention(self, name, x, ctx, heads):
        q = self.lin(name + ".to_q", x)
        kv = x if ctx is None else ctx
        k, v = self.lin(name + ".to_k", kv), self.lin(name + ".to_v", kv)
        o = ag.AttentionFn.apply(q.contiguous(), k.contiguous(), v.contiguous(), heads)
        return self.lin(name + ".to_out.0", o)


class TrainableUNet(_Graph):
    """UNet2DConditionModel (SD-v1.5 style configs, ldiffusion_amd/configs.py) with float32 master parameters in the diffusers key layout."""

    def __init__(self, cfg, state_dict, device="cuda:0"):
        super().__init__(state_dict, device, trainable=True)
        self.cfg = dict(cfg)

    def transformer(self, name, x, ctx, heads, groups):
        B, H, W, Cc = x.shape
        h = self.gn(name + ".norm", x, groups, 1e-6, False)
        h = self.conv(name + ".proj_in", h).reshape(B, H * W, Cc)
        b = name + ".transformer_blocks.0"
        h = self.attention(b + ".attn1", self.ln(b + ".norm1", h), None, heads) + h
        h = self.attention(b + ".attn2", self.ln(b + ".norm2", h), ctx, heads) + h
        f = ag.GegluFn.apply(self.lin(b + ".ff.net.0.proj", self.ln(b + ".norm3", h)).contiguous())
        h = self.lin(b + ".ff.net.2", f) + h
        return self.conv(name + ".proj_out", h.reshape(B, H, W, Cc).contiguous()) + x

    def __call__(self, sample, timestep, encoder_hidden_states):
        """sample [B, 4, h, w] float32 NCHW, timestep scalar, encoder_hidden_states [B or 1, L, cross_attention_dim] float32 (may require grad:
        the text projection is trained through it) -> eps [B, 4, h, w] float32."""
        cfg = self.cfg
        boc, groups, eps, heads, lpb = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["norm_eps"], cfg["attention_head_dim"], cfg["layers_per_block"]
        B = sample.shape[0]
        ctx = encoder_hidden_states.to(self.device)
        ctx = ctx.expand(B, -1, -1) if ctx.shape[0] != B else ctx
        ctx16 = ctx.to(torch.float16).contiguous()
        half = boc[0] // 2
        freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=self.device) / (half - cfg["freq_shift"]))
        arg = torch.full((), float(timestep), dtype=torch.float32, device=self.device) * freq   # (a fill, not a host copy: capturable)
        emb = torch.cat([torch.sin(arg), torch.cos(arg)])
        if cfg["flip_sin_to_cos"]:
            emb = torch.cat([emb[half:], emb[:half]])
        emb = emb[None].expand(B, -1).to(torch.float16).contiguous()
        temb = self.lin("time_embedding.linear_2", ag.silu(self.lin("time_embedding.linear_1", emb)))
        temb_act = ag.silu(temb)
        x = self.conv("conv_in", _nhwc16(sample.to(self.device, torch.float32)))
        skips = [x]
        for i, bt in enumerate(cfg["down_block_types"]):
            for j in range(lpb):
                x = self.resnet(f"down_blocks.{i}.resnets.{j}", x, temb_act, groups, eps)
                if bt == "CrossAttnDownBlock2D":
                    x = self.transformer(f"down_blocks.{i}.attentions.{j}", x, ctx16, heads, groups)
                skips.append(x)
            if i != len(boc) - 1:
                x = self.conv(f"down_blocks.{i}.downsamplers.0.conv", x, stride=2)
                skips.append(x)
        x = self.resnet("mid_block.resnets.0", x, temb_act, groups, eps)
        x = self.transformer("mid_block.attentions.0", x, ctx16, heads, groups)
        x = self.resnet("mid_block.resnets.1", x, temb_act, groups, eps)
        for i, bt in enumerate(cfg["up_block_types"]):
            for j in range(lpb + 1):
                x = torch.cat([x, skips.pop()], -1).contiguous()
                x = self.resnet(f"up_blocks.{i}.resnets.{j}", x, temb_act, groups, eps)
                if bt == "CrossAttnUpBlock2D":
                    x = self.transformer(f"up_blocks.{i}.attentions.{j}", x, ctx16, heads, groups)
            if i != len(boc) - 1:
                x = self.conv(f"up_blocks.{i}.upsamplers.0.conv", x, ups=1)
        x = self.gn("conv_norm_out", x, groups, eps, True)
        out = self.conv("conv_out", x)[..., :cfg["out_channels"]]
        return out.permute(0, 3, 1, 2).float()


class FrozenVAEDecoder(_Graph):
    """AutoencoderKL.decode (post_quant_conv + Decoder) with frozen parameters: gradients flow to its input only."""

    def __init__(self, cfg, state_dict, device="cuda:0"):
        sd = {k: v for k, v in state_dict.items() if k.startswith(("decoder.", "post_quant_conv."))}
        super().__init__(sd, device, trainable=False)
        self.cfg = dict(cfg)

    def mid_attention(self, name, x, groups):
        B, H, W, Cc = x.shape
        h = self.gn(name + ".group_norm", x, groups, 1e-6, False).reshape(B, H * W, Cc)
        return self.attention(name, h, None, 1).reshape(B, H, W, Cc) + x

    def __call__(self, z):
        """z [B, 4, h, w] float32 NCHW (requires grad) -> decoded image [B, 3, 8h, 8w] float32."""
        cfg = self.cfg
        groups, boc, lpb = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
        x = self.conv("post_quant_conv", _nhwc16_grad(z.to(self.device)))
        x = self.conv("decoder.conv_in", x)
        x = self.resnet("decoder.mid_block.resnets.0", x, None, groups, 1e-6)
        x = self.mid_attention("decoder.mid_block.attentions.0", x, groups)
        x = self.resnet("decoder.mid_block.resnets.1", x, None, groups, 1e-6)
        for i in range(len(boc)):
            for j in range(lpb + 1):
                x = self.resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, None, groups, 1e-6)
            if i != len(boc) - 1:
                x = self.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", x, ups=1)
        x = self.gn("decoder.conv_norm_out", x, groups, 1e-6, True)
        return self.conv("decoder.conv_out", x)[..., :cfg["out_channels"]].permute(0, 3, 1, 2).float()


def _nhwc16_grad(z):
    """NCHW float32 (may require grad) -> NHWC float16 padded to 8 channels, differentiable."""
    B, Cc, H, W = z.shape
    zp = F.pad(z.permute(0, 2, 3, 1), (0, (8 - Cc % 8) % 8))
    return zp.to(torch.float16).contiguous()


_LUMA_T = {}


def _luma_weights(device):
    key = str(device)
    if key not in _LUMA_T:
        _LUMA_T[key] = torch.tensor(LUMA, dtype=torch.float32).view(1, 3, 1, 1).to(device)
    return _LUMA_T[key]


def v5_noisy(z0, timesteps, abar, u_list=None, seed=0, offset=0):
    """The noisy latents of the V5 loop (ldiffusion.py:234-237), one per timestep: z0 + Laplace(0, sqrt(1 - abar_t)).  The noise comes from the
    given uniform draws `u_list[i]` (parity is defined given u) or from the device Philox stream (seed, offset + i * numel)."""
    from .pipeline import laplace_noise
    out = []
    for i, t in enumerate(timesteps):
        scale = float(torch.sqrt(1 - abar[int(t)]))
        out.append(laplace_noise(z0, scale, u=None if u_list is None else u_list[i], seed=seed, offset=offset + i * z0.numel()))
    return out


def v5_features(unet, vae_dec, z0, text_embeddings, timesteps, abar, u_list=None, out_hw=64, seed=0, offset=0, noisy_list=None):
    """The V5 loop (ldiffusion.py:231-247): returns (features [B, n, out_hw, out_hw] float32, last rgb).  `noisy_list`: the noisy latents
    when the caller has drawn them already (GraphedStep: the draw is outside the captured graph)."""
    if noisy_list is None:
        noisy_list = v5_noisy(z0, timesteps, abar, u_list, seed, offset)
    grays, rgb = [], None
    for i, t in enumerate(timesteps):
        den = unet(noisy_list[i], t, text_embeddings)
        rgb = F.interpolate(vae_dec(den), size=(out_hw, out_hw), mode="bilinear", align_corners=False)
        grays.append((rgb * _luma_weights(rgb.device)).sum(1, keepdim=True))
    return torch.cat(grays, 1), rgb


def contrastive_loss(features, pairs, temperature=0.5):
    """InfoNceLoss.compute_contrastive_loss (model/loss.py:44-109) for GIVEN sample triples: `pairs[b]` = list of
    (anchor_index, positive_index, [negative indices]) into the flattened H*W pixels of image b (the reference draws them with
    torch.randperm / randint; parity is defined given the draw).  Mean cross-entropy of [pos | negs] similarities / temperature."""
    B, n, H, W = features.shape
    feat = features.view(B, n, H * W).permute(0, 2, 1)
    triples = [tr for b in range(B) for tr in pairs[b]]
    if not triples:
        return features.sum() * 0.0
    K = len(triples[0][2])
    if K > 0 and all(len(tr[2]) == K for tr in triples):
        # every triple has the same number of negatives (the reference's sampler, model/loss.py:60-93): one gather + one cross-entropy
        dev = features.device
        i32 = dict(dtype=torch.int32, device=dev)
        bi = torch.tensor([b for b in range(B) for _ in pairs[b]], **i32)
        ai = torch.tensor([tr[0] for tr in triples], **i32)
        pi = torch.tensor([tr[1] for tr in triples], **i32)
        ni = torch.tensor([tr[2] for tr in triples], **i32)
        if not features.is_cuda:
            raise RuntimeError("contrastive_loss: the loss and its gradient run in ldiff_op_infonce; features must be on the GPU")
        return ag.InfoNceFn.apply(features.float(), bi, ai, pi, ni, temperature)
    total, count = features.new_zeros(()), 0
    for b in range(B):
        for a, p, negs in pairs[b]:
            anchor = feat[b, a][None]
            logits = torch.cat([anchor @ feat[b, p][None].t(), anchor @ feat[b, negs].t()], -1) / temperature
            total = total + F.cross_entropy(logits, torch.zeros(1, dtype=torch.long, device=features.device))
            count += 1
    return total / count


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None


def allreduce_gradients(params, world_size=None):
    """Replicated-parameter data parallelism: average the float32 gradients over the ranks with ONE flattened all-reduce (RCCL on
    ROCm; xGMI is point-to-point, so one large bucket per step beats many small ones).  No-op without a process group.
    The collective is unconditional and shape-stable: a parameter without a gradient on this rank (a batch that yields no sample triples
    returns a constant loss, model/loss.py:106-107) contributes zeros, so every rank enters the same all-reduce with the same bucket; the
    averaged gradient is written back to EVERY parameter."""
    dist = _dist()
    if dist is None:
        return
    params = list(params)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32) for p in params])
    dist.all_reduce(flat)
    flat /= dist.get_world_size() if world_size is None else world_size
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()


def _adamw_hip(p, g, m, v, step, lr, betas, eps, weight_decay):
    """AdamW on one contiguous float32 CUDA range (ldiff_op_adamw); the sharded optimizer's default update."""
    from . import _lib
    lib = _lib.load()
    if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()):
        raise ValueError("ShardedAdamW: the HIP update needs contiguous float32 CUDA tensors")
    _lib.check(lib.ldiff_op_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                  float(weight_decay), int(step), ag._sp()))


class ShardedAdamW:
    """ZeRO-style gradient exchange and optimizer of the fine-tuning step (the reference trains under DeepSpeed ZeRO stage 3,
    ldiffusion.py:165-193): the float32 parameters live in ONE flat buffer (the parameter tensors become views of it); per step
        reduce-scatter of the flat gradient bucket (every rank receives the sum of ITS 1/W slice: on xGMI all seven links carry 1/W
        of the bucket instead of the whole bucket per link pair),
        global-norm clipping from the slice norms (one scalar all-reduce),
        AdamW on the slice only -- the two moment buffers exist for 1/W of the parameters per rank --,
        all-gather of the updated slices straight into the flat parameter buffer.
    Without a process group the slice is the whole buffer and no collective runs.  `update` is the elementwise AdamW of a contiguous range
    (default: the HIP kernel); the CPU tests of the collective logic inject a torch formulation, the product never falls back to one.

    `partition_params=True` is ZeRO stage 3's ownership (ldiffusion.py:165-193, "stage": 3): a rank OWNS only its 1/W slice of the float32 masters
    (`master`, beside its slice of the two moments); the update writes the slice and nothing else, and the full parameters exist only between
    `gather()` -- the all-gather in front of the forward, which `train_step` / `train_step_graphed` call -- and the next update.  The flat gather
    buffer itself stays allocated (the captured step graph holds the parameters' addresses; 3.4 GB at SD-v1.5 size on a 288 GB part), so this is
    ZeRO-3's communication pattern and ownership, not its memory saving; `poison_released=True` (tests) fills the buffer with NaN after every update
    to prove that nothing reads parameters that were not gathered."""

    def __init__(self, params, lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, update=None, partition_params=False, poison_released=False):
        self.params = list(params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.update = update or _adamw_hip
        dist = _dist()
        self.world = dist.get_world_size() if dist else 1
        self.rank = dist.get_rank() if dist else 0
        n = sum(p.numel() for p in self.params)
        self.n = n
        self.shard = (n + self.world - 1) // self.world
        dev = self.params[0].device
        self.flat = torch.zeros(self.shard * self.world, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in self.params:   # the parameters become views of the flat buffer: the all-gather updates them in place
                self.flat[off:off + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self.gflat = torch.zeros_like(self.flat)
        lo = self.rank * self.shard
        self.pshard = self.flat[lo:lo + self.shard]
        self.m = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.gshard = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.skipped = 0
        self.partition_params, self.poison_released = bool(partition_params), bool(poison_released)
        self.master = self.pshard.clone() if self.partition_params else None   # this rank's slice of the masters: the only copy it keeps between steps
        self.gathered = True                                                     # the flat buffer was just built from the full parameters

    def gather(self):
        """ZeRO-3's all-gather in front of the forward (partition_params): every rank's master slice into the flat parameter buffer (the parameter
        tensors are views of it).  No-op for replicated masters, and when the buffer is already current."""
        if not self.partition_params or self.gathered:
            return
        dist = _dist()
        if dist is not None:
            dist.all_gather_into_tensor(self.flat, self.master)
        else:
            self.flat[: self.shard].copy_(self.master)
        self.gathered = True

    def step(self, max_grad_norm=None):
        """Exchange, clip, update.  Returns the global gradient norm (after averaging, before clipping); a non-finite norm skips the update."""
        dist = _dist()
        off = 0
        for p in self.params:   # shape-stable bucket: zeros where a parameter has no gradient on this rank
            if p.grad is not None:
                self.gflat[off:off + p.numel()].copy_(p.grad.reshape(-1))
            else:
                self.gflat[off:off + p.numel()].zero_()
            off += p.numel()
        if dist is not None:
            dist.reduce_scatter_tensor(self.gshard, self.gflat)
            self.gshard /= self.world
        else:
            self.gshard.copy_(self.gflat[: self.shard])
        sq = (self.gshard.double() ** 2).sum()
        if dist is not None:
            dist.all_reduce(sq)
        total = float(sq.sqrt())
        if not math.isfinite(total):
            # an overflowed float16 activation gradient (static / dynamic loss scale): every rank sees the same all-reduced norm, so every rank
            # skips -- parameters, moments and the step count stay as they are (the caller lowers the loss scale, train.finish_step)
            self.skipped += 1
            return total
        if max_grad_norm is not None:
            coef = float(max_grad_norm) / max(total, float(max_grad_norm))
            if coef < 1.0:
                self.gshard *= coef
        self.step_count += 1
        if self.partition_params:   # the owned slice only; the full parameters come back with the next gather()
            self.update(self.master, self.gshard, self.m, self.v, self.step_count, self.lr, self.betas, self.eps, self.weight_decay)
            self.gathered = False
            if self.poison_released:
                self.flat.fill_(float("nan"))
            return total
        self.update(self.pshard, self.gshard, self.m, self.v, self.step_count, self.lr, self.betas, self.eps, self.weight_decay)
        if dist is not None:
            dist.all_gather_into_tensor(self.flat, self.pshard.clone())
        return total


def clip_grad_norm(params, max_norm):
    """DeepSpeed's `gradient_clipping` (ldiffusion.py:187: 1.0): scale all gradients by max_norm / max(global L2 norm, max_norm).
    Multi-tensor reductions (688 gradient tensors at SD-v1.5 width: one norm launch each took 19 ms of a step)."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0.0
    total = float(torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads))))
    if max_norm is None or not math.isfinite(total):
        return total            # non-finite: the caller skips the update (finish_step); scaling by inf * 0 would plant NaN everywhere
    coef = float(max_norm) / max(total, float(max_norm))
    if coef < 1.0:
        torch._foreach_mul_(grads, coef)
    return total


LOSS_SCALE = 1024.0   # initial loss scale of the backward pass: activation gradients are stored in float16 (the reference trains in fp32,
                      # ldiffusion.py:172 `fp16.enabled: False`); unscaled, 0.5 % of the non-negligible parameter-gradient entries underflow to zero


LOSS_SCALE_GROWTH_INTERVAL = 200   # consecutive finite steps before a lowered loss scale doubles again (never above LOSS_SCALE)


def finish_step(params, opt_state, lr, weight_decay, max_grad_norm, optimizer=None, update=None):
    """What follows the backward pass on every rank: gradient exchange -> global norm -> (clipping) -> AdamW.  The norm is taken AFTER the
    exchange, so all ranks see the same value and take the same branch: a non-finite norm (a float16 activation gradient overflowed under the
    loss scale; the reference trains in fp32 and cannot hit this) skips the update -- parameters, moments and the AdamW step count untouched --
    and halves `opt_state["loss_scale"]` (dynamic loss scaling; the next step's backward reads it); LOSS_SCALE_GROWTH_INTERVAL finite steps in a row
    double a lowered scale again, up to LOSS_SCALE.  Returns True when the update ran.
    `optimizer` = a ShardedAdamW (reduce-scatter / sharded update / all-gather) or None (flattened all-reduce + ag.adamw_step)."""
    if optimizer is not None:
        total = optimizer.step(max_grad_norm)
    else:
        allreduce_gradients(params)
        total = clip_grad_norm(params, max_grad_norm)
    if not math.isfinite(total):
        opt_state["skipped_steps"] = opt_state.get("skipped_steps", 0) + 1
        opt_state["loss_scale"] = max(1.0, opt_state.get("loss_scale", LOSS_SCALE) * 0.5)
        opt_state["finite_steps"] = 0
        return False
    # growth rule of dynamic loss scaling: after LOSS_SCALE_GROWTH_INTERVAL consecutive finite steps the scale doubles, up to the initial
    # LOSS_SCALE -- one transient overflow must not pin the scale low for the rest of the run (the underflow LOSS_SCALE exists to avoid would
    # come back).  `total` is the exchanged norm, identical on every rank, so the decision is rank-uniform.
    opt_state["finite_steps"] = opt_state.get("finite_steps", 0) + 1
    if opt_state["finite_steps"] >= LOSS_SCALE_GROWTH_INTERVAL and opt_state.get("loss_scale", LOSS_SCALE) < LOSS_SCALE:
        opt_state["loss_scale"] = min(LOSS_SCALE, opt_state["loss_scale"] * 2.0)
        opt_state["finite_steps"] = 0
    if optimizer is None:
        (update or ag.adamw_step)(params, [p.grad for p in params], opt_state, lr=lr, weight_decay=weight_decay)
    return True


def text_projection(text_hidden, weight, bias):
    """The trainable projection of the CLIP hidden states onto the UNet's cross-attention width (segmentor.py:33-35, ldiffusion.py:142-146:
    nn.Linear(768, cross_attention_dim)) on the library's GEMM with its dgrad / wgrad (float16 operands, fp32 accumulate, float32 parameter
    gradients) -- every FLOP of the step is in libldiff_hip.so; the UNet consumes the context as float16 anyway."""
    return ag.linear(text_hidden.to(torch.float16).contiguous(), weight, bias)


def train_step(unet, vae_dec, proj, z0, text_hidden, timesteps, abar, u_list, pairs, opt_state, lr=1e-5, weight_decay=0.01, loss_fn=None,
               max_grad_norm=None, seed=0, offset=0, loss_scale=None, optimizer=None):
    """One fine-tuning step (ldiffusion.py:209-255): text projection -> V5 features -> loss -> backward through the VAE decoder and the
    UNet -> gradient all-reduce -> (clipping) -> AdamW on the UNet and projection parameters.
    `proj` = (weight [D, 768], bias [D]) float32 CUDA parameters of the text projection.  `loss_fn(features, last_rgb)` defaults to the
    contrastive loss on the given sample triples `pairs`.  Returns the loss value.  `loss_scale` None: opt_state["loss_scale"] (starts at
    LOSS_SCALE, halved by finish_step whenever the exchanged gradient norm is not finite)."""
    params = unet.parameters() + list(proj)
    if optimizer is not None:
        optimizer.gather()   # parameter partitioning (ShardedAdamW partition_params): the all-gather in front of the forward
    if loss_scale is None:
        loss_scale = opt_state.setdefault("loss_scale", LOSS_SCALE)
    for p in params:
        p.grad = None
    ctx = text_projection(text_hidden, proj[0], proj[1])
    feats, rgb = v5_features(unet, vae_dec, z0, ctx, timesteps, abar, u_list, seed=seed, offset=offset)
    loss = contrastive_loss(feats, pairs) if loss_fn is None else loss_fn(feats, rgb)
    (loss * loss_scale).backward()
    if loss_scale != 1.0:   # the parameter gradients are float32: unscale before the exchange, the clipping and AdamW
        torch._foreach_mul_([p.grad for p in params if p.grad is not None], 1.0 / loss_scale)
    finish_step(params, opt_state, lr, weight_decay, max_grad_norm, optimizer=optimizer)   # optimizer: a ShardedAdamW over `params` (same collectives as the graphed step's)
    return float(loss.detach())


class GraphedStep:
    """Forward, contrastive loss and backward of one fine-tuning step (ldiffusion.py:227-254) captured ONCE as a HIP graph
    (torch.cuda.CUDAGraph: the ~3,000 launches of the two passes -- libldiff_hip.so kernels and torch's tensor plumbing alike -- become one
    graph launch) and replayed per step.  The step driven from Python leaves the device idle 39-44 % of the time.
    What stays outside the graph: the Laplace draw (its Philox offset advances per step), the copy of the batch into the static input
    buffers, the gradient exchange, clipping and AdamW (its bias correction takes the step count as a launch argument).
    The sample triples vary in number from batch to batch: the loss launch is capacity-sized (`max_triples`) and reads the count from
    device memory (ldiff_op_infonce, T_dev)."""

    def __init__(self, unet, vae_dec, proj, batch, timesteps, abar, latent_hw=8, text_len=6, text_dim=768, out_hw=64, max_triples=1024,
                 num_negatives=1024, temperature=0.5, loss_scale=LOSS_SCALE):
        dev = unet.device
        self.unet, self.vae_dec, self.proj = unet, vae_dec, proj
        self.timesteps, self.abar = [int(t) for t in timesteps], abar.detach().cpu()
        self.out_hw, self.temperature, self.loss_scale = out_hw, temperature, float(loss_scale)
        # the scale is a DEVICE scalar inside the captured graph (finish_step halves it after an overflow: no re-capture)
        self.scale_t = torch.full((), float(loss_scale), dtype=torch.float32, device=dev)
        self.inv_scale_t = torch.full((), 1.0 / float(loss_scale), dtype=torch.float32, device=dev)
        self.noisy = [torch.zeros((batch, 4, latent_hw, latent_hw), device=dev) for _ in self.timesteps]
        self.hidden = torch.zeros((batch, text_len, text_dim), device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.bi, self.ai, self.pi = torch.zeros(max_triples, **i32), torch.zeros(max_triples, **i32), torch.zeros(max_triples, **i32)
        self.ni = torch.zeros((max_triples, num_negatives), **i32)
        self.count = torch.zeros((), **i32)
        self.params = unet.parameters() + list(proj)
        self.stream = torch.cuda.Stream(device=dev)   # warm-up and capture on ONE stream: the library keeps its scratch per (device, stream)
        self.graph, self.loss = None, None
        # every weight layout of a step in ONE launch at its start (inside the graph: the masters change with every AdamW step); the frozen
        # decoder's once, here.  Launched one by one they were 642 graph nodes of ~14 us: 9 ms of a 38 ms replay.
        self.plan_unet = ag.PackPlan([v for v in unet.p.values() if v.dim() in (2, 4)])
        self.plan_dec = ag.PackPlan([v for v in vae_dec.p.values() if v.dim() in (2, 4)])
        self.plan_dec.run()

    def _forward_backward(self):
        # The graph is built on fresh leaves that ALIAS the parameters' storage: a parameter's gradient arrives at its AccumulateGrad node on the
        # stream that node was created on, and an autograd graph the caller still holds from an eager step keeps the old nodes (default stream)
        # alive -- the engine would synchronise the capturing stream with the default stream inside the capture (hipStreamEndCapture crashes).
        own = self.unet.p
        alias = {k: (v.detach().requires_grad_(True) if v.requires_grad else v) for k, v in own.items()}
        pw, pb = (t.detach().requires_grad_(True) for t in self.proj)
        self.unet.p = alias
        try:
            self.plan_unet.run()
            with ag.packed_weights(self.plan_unet, self.plan_dec):
                ctx = text_projection(self.hidden, pw, pb)
                feats, _ = v5_features(self.unet, self.vae_dec, None, ctx, self.timesteps, self.abar, out_hw=self.out_hw, noisy_list=self.noisy)
                loss = ag.InfoNceFn.apply(feats, self.bi, self.ai, self.pi, self.ni, self.temperature, self.count)
                leaves = [v for v in alias.values() if v.requires_grad] + [pw, pb]   # the order of self.params
                grads = torch.autograd.grad(loss * self.scale_t, leaves, allow_unused=True)
        finally:
            self.unet.p = own
        live = [g for g in grads if g is not None]
        if live:
            torch._foreach_mul_(live, self.inv_scale_t)
        return loss.detach(), list(grads)

    def set_loss_scale(self, scale):
        """New loss scale for the following replays (device scalars read by the captured graph)."""
        if float(scale) != self.loss_scale:
            self.loss_scale = float(scale)
            self.scale_t.fill_(self.loss_scale)
            self.inv_scale_t.fill_(1.0 / self.loss_scale)

    def _capture(self):
        cur = torch.cuda.current_stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for _ in range(2):   # sizes every lazily grown buffer (library scratch, derived weight layouts, allocator pools) before the capture
                self._forward_backward()
        cur.wait_stream(self.stream)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.loss, self.grads = self._forward_backward()   # the graph's static loss and gradient buffers
        self.graph.replay()   # (a capture records, it does not execute)

    def set_batch(self, z0, text_hidden, pairs, u_list=None, seed=0, offset=0):
        for dst, src in zip(self.noisy, v5_noisy(z0, self.timesteps, self.abar, u_list, seed, offset)):
            dst.copy_(src)
        self.hidden.copy_(text_hidden)
        triples = [(b, tr) for b in range(len(pairs)) for tr in pairs[b]]
        T, K = len(triples), self.ni.shape[1]
        if T > self.bi.numel():
            raise ValueError(f"GraphedStep: {T} sample triples exceed max_triples = {self.bi.numel()}")
        if any(len(tr[2]) != K for _, tr in triples):
            raise ValueError(f"GraphedStep: every triple must carry num_negatives = {K} negatives")
        npix = self.out_hw * self.out_hw   # the indices address the feature map of the captured graph: validate on the host, before the upload
        if T and (len(pairs) > self.hidden.shape[0] or min(min(tr[0], tr[1], min(tr[2])) for _, tr in triples) < 0
                  or max(max(tr[0], tr[1], max(tr[2])) for _, tr in triples) >= npix):
            raise IndexError(f"GraphedStep: sample indices out of range for {self.hidden.shape[0]} feature maps of {self.out_hw}x{self.out_hw}")
        if T:
            self.bi[:T].copy_(torch.tensor([b for b, _ in triples], dtype=torch.int32))
            self.ai[:T].copy_(torch.tensor([tr[0] for _, tr in triples], dtype=torch.int32))
            self.pi[:T].copy_(torch.tensor([tr[1] for _, tr in triples], dtype=torch.int32))
            self.ni[:T].copy_(torch.tensor([tr[2] for _, tr in triples], dtype=torch.int32))
        self.count.fill_(T)

    def __call__(self, z0, text_hidden, pairs, u_list=None, seed=0, offset=0):
        """Loads the batch, replays forward + backward; the parameter gradients are in `.grad` afterwards.  Returns the loss (device scalar)."""
        self.set_batch(z0, text_hidden, pairs, u_list, seed, offset)
        if self.graph is None:
            self._capture()   # (the capture itself computes this batch)
        else:
            self.graph.replay()
        for p, g in zip(self.params, self.grads):   # (a caller may have cleared or replaced .grad since the capture)
            p.grad = g
        return self.loss


def train_step_graphed(gstep, z0, text_hidden, u_list, pairs, opt_state, lr=1e-5, weight_decay=0.01, max_grad_norm=None, seed=0, offset=0,
                       optimizer=None):
    """`train_step` with forward + backward replayed from `gstep` (GraphedStep); exchange, clipping and AdamW as in the eager step, or --
    `optimizer` = a ShardedAdamW over `gstep.params` -- reduce-scatter, sharded AdamW, all-gather.  (A ShardedAdamW moves the parameters
    into its flat buffer: create it BEFORE the GraphedStep, whose graph and weight-layout plan hold the parameters' addresses.)"""
    if optimizer is not None:
        optimizer.gather()   # parameter partitioning (ShardedAdamW partition_params): the all-gather in front of the forward
    gstep.set_loss_scale(opt_state.setdefault("loss_scale", gstep.loss_scale))
    loss = gstep(z0, text_hidden, pairs, u_list, seed, offset)
    finish_step(gstep.params, opt_state, lr, weight_decay, max_grad_norm, optimizer=optimizer)
    return float(loss)


def run_step(gstep, unet, vae_dec, proj, z0, text_hidden, timesteps, abar, pairs, opt_state, graph_ok=True, **kw):
    """The per-batch dispatch of LDiffusionModel.train_ldiffusion (ldiffusion.py:227-255: every rank runs engine.backward + engine.step for EVERY
    batch): the graph replay where the batch fits the capture, the eager step otherwise -- and NEVER nothing: a rank whose batch yields no
    sample triples still runs the eager step (its loss is a constant 0, its gradients are zeros), so that it enters the same gradient
    collective as the other ranks and AdamW advances in lock-step everywhere.  Returns (loss value, "graph" | "eager")."""
    n_tr = sum(len(t) for t in pairs)
    B = z0.shape[0]
    fits = (graph_ok and gstep is not None and n_tr > 0 and B == gstep.hidden.shape[0] and [int(t) for t in timesteps] == gstep.timesteps
            and n_tr <= gstep.bi.numel() and tuple(z0.shape[-2:]) == tuple(gstep.noisy[0].shape[-2:]) and text_hidden.shape[1] == gstep.hidden.shape[1])
    if fits:
        return train_step_graphed(gstep, z0, text_hidden, None, pairs, opt_state, **kw), "graph"
    return train_step(unet, vae_dec, proj, z0, text_hidden, timesteps, abar, None, pairs, opt_state, **kw), "eager"
