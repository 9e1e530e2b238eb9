"""-m gpu: the fine-tuning step (ldiffusion_amd/train.py; /root/reference/ldiffusion.py:209-255) -- forward AND backward on the HIP
kernels -- against torch.autograd over the CPU oracle's plain-torch graphs (oracle/unet.py, oracle/vae.py) at reduced width and the
reference's training size (64 x 64 images = 8 x 8 latents).

Tolerance: float16 activations / activation gradients with float32 accumulation through ~500 chained forward and ~1000 backward ops:
the features within 1e-2 of their range (all-fp16 storage here: the inference path's split residual stream is not used in training), the
loss within 2e-3, every parameter gradient within 5e-2 of its largest reference entry (tensors whose gradient is mathematically zero --
q/k projections of the 1-token self-attention at the 1x1 level -- are measured against 1e-3 of the largest gradient entry of the model),
and the direction of the whole gradient (cosine over all 688 parameter tensors) >= 0.9995.  Measured values are printed."""
import pytest
import torch
import torch.nn.functional as F

from ldiffusion_amd import configs, train, weights
from ldiffusion_amd.loss import InfoNceLoss
from oracle import noise_post, schedule, unet as ounet, vae as ovae

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,n,hw,K,per", [(2, 5, 64, 1024, 12), (1, 1, 16, 7, 3), (3, 32, 32, 300, 5)])
def test_contrastive_loss_kernel_value_and_gradient(B, n, hw, K, per):
    """ldiff_op_infonce (one launch: loss + d loss / d features) against the host statement of model/loss.py:89-109 in float64 for the
    same triples: the reference's shape (5 V5 planes, 1024 negatives), the smallest one, and the plane limit.  <= 1e-5 relative."""
    g = torch.Generator().manual_seed(5)
    feats = torch.rand((B, n, hw, hw), generator=g) * 2 - 0.5
    pairs = [[(int(torch.randint(0, hw * hw, (1,), generator=g)), int(torch.randint(0, hw * hw, (1,), generator=g)),
               torch.randperm(hw * hw, generator=g)[:K].tolist()) for _ in range(per)] for _ in range(B)]
    pairs[-1] = pairs[-1][:1]                                   # images contribute different numbers of triples
    ref_in = feats.double().requires_grad_(True)
    ref = InfoNceLoss().compute_contrastive_loss(ref_in, None, triples=pairs)
    ref.backward()
    x = feats.to(DEV).requires_grad_(True)
    loss = train.contrastive_loss(x, pairs)
    (loss * 3.0).backward()
    e_l = abs(loss.item() - ref.item()) / abs(ref.item())
    e_g = ((x.grad.cpu().double() / 3.0 - ref_in.grad).abs().max() / ref_in.grad.abs().max()).item()
    print(f"\n[infonce B={B} n={n} {hw}x{hw} K={K}] loss {loss.item():.6f} (ref {ref.item():.6f}, rel {e_l:.1e}); gradient max err / max {e_g:.1e}")
    assert e_l <= 1e-5 and e_g <= 1e-5
    again = train.contrastive_loss(feats.to(DEV), pairs)        # the outputs are overwritten, not accumulated
    assert abs(again.item() - loss.item()) <= 1e-6 * abs(loss.item())


def _setup(sd15=False):
    """sd15: the graphs of BASELINE.json configs[4] at their real width (859.5 M trainable UNet parameters, 768 -> 768 text projection),
    the reference's training size (8 x 8 latents) and its one V5 pass per step (num_inference_steps 5 // 5, ldiffusion.py:198)."""
    ucfg, vcfg = (configs.SD15_UNET, configs.SD15_VAE) if sd15 else (configs.TINY_UNET, configs.TINY_VAE)
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    g = torch.Generator().manual_seed(77)
    B = 2
    D, Dh = ucfg["cross_attention_dim"], (768 if sd15 else 32)
    z0 = torch.randn((B, 4, 8, 8), generator=g) * 0.8
    hidden = torch.randn((1, 6, Dh), generator=g) * 0.5                                   # text_encoder(ids).last_hidden_state stand-in
    proj_w = (torch.randn((D, Dh), generator=g) / Dh ** 0.5).to(torch.float16).float()
    proj_b = (torch.randn(D, generator=g) * 0.05).to(torch.float16).float()
    sch = schedule.PNDMOracle()
    sch.set_timesteps(1 if sd15 else 2)
    ts = [int(t) for t in sch.timesteps]
    eps32 = torch.finfo(torch.float32).eps
    u_list = [torch.rand((B, 4, 8, 8), generator=g) * (2 - eps32) + (eps32 - 1) for _ in ts]
    pairs = []
    for b in range(B):                                                                    # (anchor, positive, negatives) pixel indices of a 64x64 map
        pairs.append([(int(torch.randint(0, 4096, (1,), generator=g)), int(torch.randint(0, 4096, (1,), generator=g)),
                       torch.randint(0, 4096, (64,), generator=g).tolist()) for _ in range(6)])
    return ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs


def _oracle_loss_and_grads(ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs):
    sd = {k: v.clone().requires_grad_(True) for k, v in usd.items()}
    pw, pb = proj_w.clone().requires_grad_(True), proj_b.clone().requires_grad_(True)
    ctx = F.linear(hidden, pw, pb)
    grays = []
    for i, t in enumerate(ts):
        noisy = noise_post.laplace_forward_noise(z0, sch.alphas_cumprod[t], u_list[i])
        den = ounet.unet_forward(sd, ucfg, noisy, t, ctx).sample
        rgb = F.interpolate(ovae.vae_decode(vsd, vcfg, den), size=(64, 64), mode="bilinear", align_corners=False)
        grays.append((rgb * torch.tensor(train.LUMA).view(1, 3, 1, 1)).sum(1, keepdim=True))
    feats = torch.cat(grays, 1)
    loss = InfoNceLoss().compute_contrastive_loss(feats, None, triples=pairs)   # CPU statement, pinned to model/loss.py by tests/test_cpu_oracle.py
    loss.backward()
    return loss.detach(), feats.detach(), {k: v.grad for k, v in sd.items()}, pw.grad, pb.grad


def _compare_step(sd15, tol_feat, tol_loss, tol_grad, min_cos):
    ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs = _setup(sd15)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    rloss, rfeats, rgrads, rpw, rpb = _oracle_loss_and_grads(ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs)
    unet = train.TrainableUNet(ucfg, usd, DEV)
    dec = train.FrozenVAEDecoder(vcfg, vsd, DEV)
    pw, pb = proj_w.to(DEV).requires_grad_(True), proj_b.to(DEV).requires_grad_(True)
    ctx = train.text_projection(hidden.to(DEV), pw, pb)   # the product path: the library's GEMM (the oracle side above is F.linear in fp32)
    feats, _ = train.v5_features(unet, dec, z0.to(DEV), ctx, ts, sch.alphas_cumprod, [u.to(DEV) for u in u_list])
    loss = train.contrastive_loss(feats, pairs)
    (loss * train.LOSS_SCALE).backward()                       # as train_step does: float16 activation gradients behind a static loss scale
    for p_ in list(unet.p.values()) + [pw, pb]:
        if p_.grad is not None:
            p_.grad.mul_(1.0 / train.LOSS_SCALE)
    e_f = ((feats.detach().cpu() - rfeats).abs().max() / rfeats.abs().max()).item()
    e_l = abs(loss.item() - rloss.item()) / abs(rloss.item())
    worst, dots, n1, n2 = (0.0, ""), 0.0, 0.0, 0.0
    gmax = max(float(v.abs().max()) for v in rgrads.values())
    missing = [k for k, p in unet.p.items() if p.grad is None]
    assert not missing, f"no gradient reached {missing[:5]}"
    lost_where = []
    lost, nonzero = 0, 0   # float16 activation gradients without loss scaling: entries that came out exactly 0 where the reference's are not small
    for k, p in list(unet.p.items()) + [("proj.weight", pw), ("proj.bias", pb)]:
        ref = rgrads[k] if k in rgrads else (rpw if k == "proj.weight" else rpb)
        got = p.grad.detach().cpu()
        e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-3 * gmax)).item()
        if e > worst[0]:
            worst = (e, k)
        big = ref.abs() > 1e-3 * ref.abs().max().clamp_min(1e-3 * gmax)   # (tensors whose gradient is mathematically 0 hold only rounding noise)
        nl = int(((got == 0) & big).sum())
        lost += nl; nonzero += int(big.sum())
        if nl:
            lost_where.append((nl, k, tuple(ref.shape)))
        dots += (got.double() * ref.double()).sum().item(); n1 += got.double().pow(2).sum().item(); n2 += ref.double().pow(2).sum().item()
    cos = dots / (n1 ** 0.5 * n2 ** 0.5)
    print(f"training step ({'SD-v1.5' if sd15 else 'tiny'} width, 8x8 latents, {len(ts)} V5 pass(es)): features {e_f:.2e}, loss {loss.item():.5f} vs {rloss.item():.5f} ({e_l:.2e}); "
          f"{len(unet.p) + 2} parameter gradients: worst {worst[0]:.2e} ({worst[1]}), cosine {cos:.6f}; entries lost to fp16 underflow {lost} of {nonzero} ({lost / max(nonzero, 1):.2e})")
    print("  exact zeros by tensor:", sorted(lost_where, reverse=True)[:8])
    assert e_f <= tol_feat and e_l <= tol_loss and worst[0] <= tol_grad and cos >= min_cos
    assert lost <= 1e-4 * nonzero, "float16 activation gradients underflow: a loss scale is needed at this width"
    assert all(not p.requires_grad and p.grad is None for p in dec.p.values())            # the VAE is frozen: no parameter gradients


@pytest.mark.timeout(900)
def test_training_step_gradients_match_autograd_over_the_oracle():
    _compare_step(False, 1e-2, 2e-3, 2.5e-2, 0.9995)     # measured: features 3e-3, loss 9e-5, worst gradient 1.4e-2, cosine 0.99998


@pytest.mark.timeout(3000)
def test_training_step_at_sd15_width_matches_autograd_over_the_oracle():
    """BASELINE.json configs[4] at SD-v1.5 width (fp16 operands, one rank): B = 2, 8 x 8 latents, one V5 pass, 860 M parameter gradients
    against torch.autograd over the fp32 CPU oracle."""
    _compare_step(True, 1e-2, 2e-3, 4e-2, 0.9995)        # measured: features 5.5e-3, loss 7e-5, worst gradient 2.2e-2, cosine 0.99993


def test_train_step_runs_and_updates_parameters():
    ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs = _setup()
    unet = train.TrainableUNet(ucfg, usd, DEV)
    dec = train.FrozenVAEDecoder(vcfg, vsd, DEV)
    proj = (proj_w.to(DEV).requires_grad_(True), proj_b.to(DEV).requires_grad_(True))
    before = {k: v.detach().clone() for k, v in list(unet.p.items())[:5]}
    state = {}
    losses = [train.train_step(unet, dec, proj, z0.to(DEV), hidden.to(DEV), ts, sch.alphas_cumprod, [u.to(DEV) for u in u_list], pairs, state, lr=1e-4)
              for _ in range(3)]
    print(f"three AdamW steps on one batch: loss {[round(x, 5) for x in losses]}")
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]
    assert state["step"] == 3 and all(not torch.equal(unet.p[k].detach(), v) for k, v in before.items())


def test_graphed_step_equals_the_eager_step():
    """train.GraphedStep (forward + loss + backward captured once as a HIP graph, replayed per step) against the eager step on the same
    batches: the first call captures, the second replays on a DIFFERENT batch with a different number of sample triples (the loss launch is
    capacity-sized and reads the count from device memory).  Same kernels, same order: loss and every parameter gradient agree to float
    atomics' reordering (the contrastive loss scatters its gradient with atomic adds)."""
    ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs = _setup()
    unet = train.TrainableUNet(ucfg, usd, DEV)
    dec = train.FrozenVAEDecoder(vcfg, vsd, DEV)
    pw, pb = proj_w.to(DEV).requires_grad_(True), proj_b.to(DEV).requires_grad_(True)
    params = unet.parameters() + [pw, pb]
    g = torch.Generator().manual_seed(3)
    batch2 = (z0 * 0.7 + 0.1, hidden.flip(0) * 1.1, [u.flip(0) for u in u_list],
              [[(int(torch.randint(0, 4096, (1,), generator=g)), int(torch.randint(0, 4096, (1,), generator=g)), torch.randint(0, 4096, (64,), generator=g).tolist())
                for _ in range(n)] for n in (3, 7)])
    gstep = train.GraphedStep(unet, dec, (pw, pb), z0.shape[0], ts, sch.alphas_cumprod, latent_hw=8, text_len=hidden.shape[1], text_dim=hidden.shape[2],
                              max_triples=32, num_negatives=64)
    for name, (z, h, ul, pr) in (("capture", (z0, hidden, u_list, pairs)), ("replay", batch2)):
        for p_ in params:
            p_.grad = None
        ctx = train.text_projection(h.to(DEV), pw, pb)
        feats, _ = train.v5_features(unet, dec, z.to(DEV), ctx, ts, sch.alphas_cumprod, [u.to(DEV) for u in ul])
        loss = train.contrastive_loss(feats, pr)
        (loss * train.LOSS_SCALE).backward()
        ref = [None if p_.grad is None else (p_.grad / train.LOSS_SCALE).clone() for p_ in params]
        got_loss = gstep(z.to(DEV), h.to(DEV), pr, [u.to(DEV) for u in ul])
        torch.cuda.synchronize()
        e_l = abs(got_loss.item() - loss.item()) / abs(loss.item())
        worst, where = 0.0, -1
        for i_, (p_, r) in enumerate(zip(params, ref)):
            assert (p_.grad is None) == (r is None)
            if r is not None:
                e_ = ((p_.grad - r).abs().max() / r.abs().max().clamp_min(1e-20)).item()
                if e_ > worst:
                    worst, where = e_, i_
        print(f"graphed step ({name}): loss {got_loss.item():.6f} vs eager {loss.item():.6f} ({e_l:.1e}); worst parameter-gradient difference {worst:.1e} of the tensor's max "
              f"(parameter {where} of {len(params)}, shape {tuple(params[where].shape)})")
        # (the loss kernel's atomic adds reorder between runs; behind them every activation gradient is float16, so a reordered sum flips roundings downstream --
        # since round 6 also in the text projection's own wgrad, which runs on the library's float16 GEMM instead of ATen's float32 one)
        assert e_l <= 1e-5 and worst <= 3e-3
    state = {}
    l0 = train.train_step_graphed(gstep, z0.to(DEV), hidden.to(DEV), [u.to(DEV) for u in u_list], pairs, state, lr=1e-4)
    l1 = train.train_step_graphed(gstep, z0.to(DEV), hidden.to(DEV), [u.to(DEV) for u in u_list], pairs, state, lr=1e-4)
    print(f"two graphed AdamW steps: loss {l0:.5f} -> {l1:.5f}, step count {state['step']}")
    assert state["step"] == 2 and l1 < l0


def test_contrastive_loss_kernel_at_the_reference_fixture():
    """tests/golden/reference_infonce.npz holds losses computed by the REFERENCE's own InfoNceLoss (model/loss.py, run in the build container by
    scripts/gen_golden_loss.py) for seeded label maps: with the same torch seed the mirror draws the same triples on the host, and the loss
    + gradient over them come from the one-launch kernel.  Value against the reference's number, gradient against the host statement."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_infonce.npz"))
    for tag in "abc":
        feats, labels = torch.from_numpy(z[tag + "_features"]), torch.from_numpy(z[tag + "_labels"])
        torch.manual_seed(int(z[tag + "_seed"]))
        triples = InfoNceLoss().sample_triples(labels)
        ref_in = feats.double().requires_grad_(True)
        InfoNceLoss().compute_contrastive_loss(ref_in, labels, triples=triples).backward()
        x = feats.to(DEV).requires_grad_(True)
        torch.manual_seed(int(z[tag + "_seed"]))
        loss = InfoNceLoss().compute_contrastive_loss(x, labels)      # draws the triples itself (host RNG), then ldiff_op_infonce
        loss.backward()
        e_l = abs(loss.item() - float(z[tag + "_loss"])) / abs(float(z[tag + "_loss"]))
        e_g = ((x.grad.cpu().double() - ref_in.grad).abs().max() / ref_in.grad.abs().max()).item()
        print(f"[infonce fixture {tag}] {sum(len(t) for t in triples)} triples: loss {loss.item():.6f} vs the reference's {float(z[tag + '_loss']):.6f} ({e_l:.1e}); gradient {e_g:.1e}")
        assert e_l <= 1e-5 and e_g <= 1e-5


def test_graphed_step_with_the_sharded_optimizer():
    """train_step_graphed(optimizer=ShardedAdamW) on one rank (the slice is the whole flat buffer, no collective): the parameters after two
    steps equal those of the multi-tensor AdamW path up to the rounding of the clipping norm (float64 over the flat slice there, float32
    per tensor here: the clip coefficient differs in its last bits; the elementwise update is the same, and the collectives of the sharded
    path are pinned bit for bit against the all-reduce path by the world-size-2 gloo test)."""
    ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs = _setup()
    dec = train.FrozenVAEDecoder(vcfg, vsd, DEV)
    finals = []
    for sharded in (False, True):
        unet = train.TrainableUNet(ucfg, usd, DEV)
        proj = (proj_w.to(DEV).requires_grad_(True), proj_b.to(DEV).requires_grad_(True))
        opt = train.ShardedAdamW(unet.parameters() + list(proj), lr=1e-4, weight_decay=0.01) if sharded else None   # before the capture
        gstep = train.GraphedStep(unet, dec, proj, z0.shape[0], ts, sch.alphas_cumprod, latent_hw=8, text_len=hidden.shape[1], text_dim=hidden.shape[2],
                                  max_triples=32, num_negatives=64)
        state, losses = {}, []
        for _ in range(2):
            losses.append(train.train_step_graphed(gstep, z0.to(DEV), hidden.to(DEV), [u.to(DEV) for u in u_list], pairs, state, lr=1e-4, weight_decay=0.01,
                                                   max_grad_norm=1.0, optimizer=opt))
        torch.cuda.synchronize()
        finals.append(([p_.detach().clone() for p_ in unet.parameters() + list(proj)], losses))
    print(f"two graphed steps: multi-tensor AdamW losses {finals[0][1]}, sharded {finals[1][1]}")
    assert finals[0][1][1] < finals[0][1][0] and abs(finals[0][1][1] - finals[1][1][1]) <= 1e-4 * abs(finals[0][1][1])
    # Parameters: Adam turns a gradient into a step of about lr whatever its size, so entries whose gradient is rounding noise (the q / k
    # projections of the 1-token attention; the loss scatters with float atomics, whose order differs from run to run) may move by lr per step
    # in either direction: the bound is 2 lr per step, and the entries that moved apart by more than a tenth of lr are a small minority.
    diffs = torch.cat([(a - b).abs().reshape(-1) for a, b in zip(finals[0][0], finals[1][0])])
    print(f"  parameters after two steps: max |difference| {diffs.max().item():.1e} (lr 1e-4), fraction beyond lr / 10: {(diffs > 1e-5).float().mean().item():.1e}")
    assert diffs.max().item() <= 4.1e-4 and (diffs > 1e-5).float().mean().item() <= 0.02


_TWO_RANK_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import test_gpu_train as T
from ldiffusion_amd import train
dist.init_process_group("gloo")                      # two ranks share cuda:0 (one-GPU box): gloo moves the CUDA buffers; on a node it is RCCL
rank, DEV = dist.get_rank(), "cuda:0"
ucfg, vcfg, usd, vsd, z0, hidden, proj_w, proj_b, sch, ts, u_list, pairs = T._setup()
unet = train.TrainableUNet(ucfg, usd, DEV)
dec = train.FrozenVAEDecoder(vcfg, vsd, DEV)
proj = (proj_w.to(DEV).requires_grad_(True), proj_b.to(DEV).requires_grad_(True))
params = unet.parameters() + list(proj)
PART = os.environ.get("LDIFF_TEST_PARTITION") == "1"                   # ZeRO-3 ownership: masters owned by slice, all-gather in front of every forward
opt = train.ShardedAdamW(params, lr=1e-4, weight_decay=0.01, partition_params=PART, poison_released=PART)   # before the capture: the parameters move into its flat buffer
assert opt.world == 2 and opt.m.numel() == (sum(p.numel() for p in params) + 1) // 2
assert (opt.master is not None and opt.master.numel() == opt.m.numel()) if PART else opt.master is None
gstep = train.GraphedStep(unet, dec, proj, z0.shape[0], ts, sch.alphas_cumprod, latent_hw=8, text_len=hidden.shape[1], text_dim=hidden.shape[2],
                          max_triples=32, num_negatives=64)
zr = (z0 * (1.0 + 0.3 * rank)).to(DEV)                                   # every rank its own batch
before = [p.detach().clone() for p in params[:4]]
state, kinds, losses = {{}}, [], []
for it in range(3):
    # step 1: rank 1's batch is all background -- no sample triples: it must still enter the reduce-scatter / all-gather (eager step, zero gradients)
    pr = [[], []] if (rank == 1 and it == 1) else pairs
    val, kind = train.run_step(gstep, unet, dec, proj, zr, hidden.to(DEV), ts, sch.alphas_cumprod, pr, state, lr=1e-4, weight_decay=0.01, max_grad_norm=1.0,
                               seed=rank, offset=it * 1000, optimizer=opt)
    kinds.append(kind); losses.append(val)
torch.cuda.synchronize()
assert kinds == (["graph", "eager", "graph"] if rank == 1 else ["graph"] * 3), (rank, kinds)
assert opt.step_count == 3 and all(l == l for l in losses)
if PART:
    assert torch.isnan(opt.flat).all()                                   # released and poisoned after the last update: only the owned slices persist
    opt.gather()
assert all(not torch.equal(p.detach(), b) for p, b in zip(params[:4], before))
flat = opt.flat.detach().to("cpu")
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
assert torch.equal(other[0], other[1]), "the ranks hold different parameters after the all-gather"
assert torch.isfinite(flat).all()
dist.barrier()
dist.destroy_process_group()
open(os.path.join({out!r}, "train%d.ok" % rank), "w").write(repr(losses))
"""


@pytest.mark.timeout(900)
@pytest.mark.parametrize("partition", [False, True])
def test_graphed_step_and_sharded_adamw_on_two_ranks(tmp_path, partition):
    """BASELINE configs[4]'s step on TWO ranks: GraphedStep (forward + loss + backward replayed from one HIP graph) + ShardedAdamW (reduce-scatter of
    the flat gradient bucket, AdamW on the rank's half with half of the moments, all-gather of the parameters) per rank, every rank its own
    batch, one rank with an all-background batch in the middle (it steps eagerly with zero gradients and must meet the same collectives).  Both
    ranks share the box's one GPU over gloo (CUDA tensors; probed: profiles/r04_gloo_cuda_probe.txt) -- on a node the same code runs over RCCL.
    Asserted: no hang, three optimizer steps everywhere, bit-identical parameters on both ranks afterwards.
    partition = True: the ZeRO-3 form the reference's DeepSpeed config asks for (ldiffusion.py:165-193, stage 3) -- each rank owns its half of the float32
    masters, the full parameters are all-gathered in front of every forward (graph replay and eager step alike), and the buffer is poisoned with NaN after
    every update: a forward that read un-gathered parameters would produce a NaN loss."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "two_rank_worker.py"
    script.write_text(_TWO_RANK_WORKER.format(root=root, out=str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               LDIFF_TEST_PARTITION="1" if partition else "0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (tmp_path / "train0.ok").exists() and (tmp_path / "train1.ok").exists(), r.stdout[-2000:] + r.stderr[-2000:]
    print("two-rank losses:", (tmp_path / "train0.ok").read_text(), (tmp_path / "train1.ok").read_text())
