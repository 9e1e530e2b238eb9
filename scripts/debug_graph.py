"""diagnostic: GraphedStep capture at a given width / number of V5 passes / batch.  usage: python scripts/debug_graph.py tiny|sd15 n_passes batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ldiffusion_amd import configs, train, weights
from ldiffusion_amd.scheduler import PNDMScheduler
DEV = "cuda:0"
which, npass, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ucfg, vcfg = (configs.SD15_UNET, configs.SD15_VAE) if which == "sd15" else (configs.TINY_UNET, configs.TINY_VAE)
usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
unet, dec = train.TrainableUNet(ucfg, usd, DEV), train.FrozenVAEDecoder(vcfg, vsd, DEV)
g = torch.Generator().manual_seed(0)
D, Dh = ucfg["cross_attention_dim"], (768 if which == "sd15" else 32)
z0 = (torch.randn((B, 4, 8, 8), generator=g) * 0.8).to(DEV)
hidden = (torch.randn((B, 6, Dh), generator=g) * 0.5).to(DEV)
proj = ((torch.randn((D, Dh), generator=g) / Dh ** 0.5).to(DEV).requires_grad_(True), torch.zeros(D, device=DEV, requires_grad=True))
sch = PNDMScheduler(); sch.set_timesteps(max(npass - 1, 1), device=DEV)
ts = [int(t) for t in sch.timesteps][:npass]
eps32 = torch.finfo(torch.float32).eps
u_list = [(torch.rand((B, 4, 8, 8), generator=g) * (2 - eps32) + (eps32 - 1)).to(DEV) for _ in ts]
pairs = [[(int(torch.randint(0, 4096, (1,), generator=g)), int(torch.randint(0, 4096, (1,), generator=g)), torch.randint(0, 4096, (64,), generator=g).tolist())
          for _ in range(4)] for _ in range(B)]
if "--hidden1" in sys.argv:
    hidden = hidden[:1].contiguous()
if "--keep-eager" in sys.argv:
    import torch.nn.functional as F
    params = unet.parameters() + list(proj)
    ctx = F.linear(hidden, proj[0], proj[1])
    feats, _ = train.v5_features(unet, dec, z0, ctx, ts, sch.alphas_cumprod, u_list)
    loss = train.contrastive_loss(feats, pairs)
    (loss * train.LOSS_SCALE).backward()
    ref = [None if p_.grad is None else (p_.grad / train.LOSS_SCALE).clone() for p_ in params]
    for p_ in params:
        p_.grad = None
if "--eager-first" in sys.argv:
    state0 = {}
    train.train_step(unet, dec, proj, z0, hidden, ts, sch.alphas_cumprod, u_list, pairs, state0, lr=0.0)
gs = train.GraphedStep(unet, dec, proj, B, ts, sch.alphas_cumprod, latent_hw=8, text_len=6, text_dim=Dh, max_triples=64, num_negatives=64)
print(which, npass, B, "timesteps", ts, flush=True)
l = gs(z0, hidden, pairs, u_list); torch.cuda.synchronize(); print("capture ok, loss", float(l), flush=True)
l = gs(z0, hidden, pairs, u_list); torch.cuda.synchronize(); print("replay ok, loss", float(l), flush=True)
if "--check" in sys.argv:
    import torch.nn.functional as F
    names = [k for k, v in unet.p.items() if v.requires_grad] + ["proj.w", "proj.b"]
    params = unet.parameters() + list(proj)
    def eager():
        for p_ in params: p_.grad = None
        ctx = F.linear(hidden, proj[0], proj[1])
        feats, _ = train.v5_features(unet, dec, z0, ctx, ts, sch.alphas_cumprod, u_list)
        loss = train.contrastive_loss(feats, pairs)
        (loss * train.LOSS_SCALE).backward()
        return [None if p_.grad is None else (p_.grad / train.LOSS_SCALE).clone() for p_ in params]
    def graph():
        if '--replay-only' in sys.argv: gs.graph.replay()
        else: gs(z0, hidden, pairs, u_list)
        torch.cuda.synchronize()
        return [None if g is None else g.clone() for g in gs.grads]
    def diff(a, b, what):
        bad = []
        for n, x, y in zip(names, a, b):
            if x is None or y is None: continue
            e = ((x - y).abs().max() / y.abs().max().clamp_min(1e-20)).item()
            if not (e <= 1e-3): bad.append((n, e, tuple(x.shape)))
        print(what, "mismatching tensors:", bad[:6], flush=True)
    e1 = eager(); e2 = eager(); diff(e1, e2, "eager vs eager")
    for rep in range(6):
        g1 = graph(); diff(g1, e1, f"graph replay {rep} vs eager")
        print('   loss', float(gs.loss), 'nan tensors', sum(int(torch.isnan(g).any()) for g in g1 if g is not None), 'of', len(g1), 'first finite:', [n for n, g in zip(names, g1) if g is not None and not torch.isnan(g).any()][:5], flush=True)
        if rep % 2: e3 = eager(); diff(e3, e1, f"eager again {rep} vs eager")
if "--poison" in sys.argv:
    def stat():
        torch.cuda.synchronize()
        return float(gs.loss), sum(int(torch.isnan(g).any()) for g in gs.grads if g is not None), float(sum(g.double().abs().sum() for g in gs.grads if g is not None))
    gs.graph.replay(); print("replay A", stat(), flush=True)
    gs.graph.replay(); print("replay B (nothing in between)", stat(), flush=True)
    junk = [torch.full((64 * 1024 * 1024,), float("nan"), device=DEV) for _ in range(16)]   # 4 GiB of NaN from the regular pool
    torch.cuda.synchronize(); del junk
    gs.graph.replay(); print("replay C (after 4 GiB of NaN allocated and freed)", stat(), flush=True)
    junk = [torch.full((64 * 1024 * 1024,), float("nan"), device=DEV) for _ in range(16)]
    gs.graph.replay(); print("replay D (4 GiB of NaN alive)", stat(), flush=True)
    del junk
    print(torch.cuda.memory_summary(abbreviated=True)[:1500])
