"""CPU (-m "not gpu"): the C-ABI library loads and exports every declared symbol; host-only entry points; patch
sharding + the N>1 gather path on gloo with world_size 2.  No compute kernel is launched here (there is no GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from ldiffusion_amd import _lib, parallel
from oracle import schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_declared_in_header(lib):
    hdr = open(os.path.join(ROOT, "include", "ldiff.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ldiff_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    assert declared == set(_lib.SIGNATURES), f"header vs ctypes table differ: {declared ^ set(_lib.SIGNATURES)}"
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libldiff_hip.so"
    assert lib.ldiff_version() == int(re.search(r"#define LDIFF_VERSION (\d+)", open(os.path.join(ROOT, "include", "ldiff.h")).read()).group(1))


def test_host_only_entry_points_match_oracle(lib):
    buf = (C.c_float * 1000)()
    assert lib.ldiff_pndm_alphas_cumprod(buf, 1000) == 0
    a, b = np.array(buf[:], np.float32), schedule.alphas_cumprod().numpy()
    assert np.abs(a - b).max() <= 4e-7            # ~2 ulp (ATen's vectorised linspace rounds twice); python passes torch's table
    ts = (C.c_int64 * 64)()
    for n_passes in (1, 3, 5, 20):
        k = lib.ldiff_plms_timesteps(n_passes, ts, 64)
        assert list(ts[:k]) == schedule.plms_timesteps(1 if n_passes == 1 else n_passes - 1).tolist() and k == n_passes
    for bad in (0, 2, -3, 1002):
        assert lib.ldiff_plms_timesteps(bad, ts, 64) == -1 and b"n_passes" in lib.ldiff_last_error()
    assert lib.ldiff_plms_timesteps(5, ts, 2) == -1   # capacity too small
    # _get_prev_sample coefficients vs the oracle's float32 torch arithmetic
    sch = schedule.PNDMOracle()
    sc, ce = C.c_float(), C.c_float()
    for t, p in [(751, 501), (501, 251), (251, 1), (1, -249)]:
        a_prev = sch.alphas_cumprod[p] if p >= 0 else sch.final_alpha_cumprod
        assert lib.ldiff_pndm_coeffs(float(sch.alphas_cumprod[t]), float(a_prev), C.byref(sc), C.byref(ce)) == 0
        osc, oce, _, _ = sch.prev_sample_coeffs(t, p)
        assert abs(sc.value - float(osc)) <= 3e-7 * abs(float(osc)) and abs(ce.value + float(oce)) <= 3e-7 * abs(float(oce))
    assert lib.ldiff_pndm_coeffs(0.0, 0.5, C.byref(sc), C.byref(ce)) == -1


def test_error_mapping_and_no_cpu_fallback(lib):
    with pytest.raises(ValueError):
        _lib.check(lib.ldiff_plms_timesteps(2, (C.c_int64 * 4)(), 4))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            _lib.require_gpu()
        from ldiffusion_amd import configs, weights
        from ldiffusion_amd.models import UNet2DConditionModel
        with pytest.raises(RuntimeError):
            UNet2DConditionModel(configs.TINY_UNET, {}, "cuda:0")   # product path refuses to run without the GPU


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ldiffusion_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        parallel.shard_range(8, 2, 2)
    m = torch.arange(6, dtype=torch.uint8).view(6, 1, 1)
    assert parallel.gather_masks(m, 6) is m          # no process group: identity on the full batch
    with pytest.raises(ValueError):
        parallel.gather_masks(m, 7)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from ldiffusion_amd import parallel
dist.init_process_group("gloo")
rank, world, _ = parallel.world_info()
assert (rank, world) == (dist.get_rank(), dist.get_world_size()) and world == 2
for total in (8, 5):                                   # even and ragged shards
    lo, hi = parallel.shard_range(total, rank, world)
    full = (torch.arange(total * 4 * 4, dtype=torch.int64) % 251).to(torch.uint8).view(total, 4, 4)   # "masks" every rank can rebuild
    got = parallel.gather_masks(full[lo:hi].clone(), total)
    assert got.dtype == torch.uint8 and torch.equal(got, full), (rank, total)
feats = torch.full((4, 5, 2, 2), rank, dtype=torch.uint8)          # features [n_r, N, H, W]
allf = parallel.gather_masks(feats, 8)
assert allf.shape == (8, 5, 2, 2) and allf[:4].eq(0).all() and allf[4:].eq(1).all()
try:
    parallel.gather_masks(torch.zeros((3, 4, 4), dtype=torch.uint8), 8)
    raise SystemExit("expected ValueError")
except ValueError:
    pass
dist.barrier()
dist.destroy_process_group()
open(os.path.join({out!r}, "rank%d.ok" % rank), "w").write("ok")   # one file per rank: the two stdouts interleave
"""


def test_gather_masks_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists(), r.stdout[-2000:] + r.stderr[-2000:]


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


_GRAD_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from ldiffusion_amd import train
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ps = [torch.nn.Parameter(torch.zeros(s)) for s in ((5,), (3, 4), (1,))]
for i, p in enumerate(ps[:2]):
    p.grad = torch.full(p.shape, float(rank + 1 + i))          # the third parameter has no gradient (unused in the step)
train.allreduce_gradients(ps)
for i, p in enumerate(ps[:2]):
    assert torch.equal(p.grad, torch.full(p.shape, (1 + 2) / 2 + i)), (rank, i, p.grad)
assert torch.equal(ps[2].grad, torch.zeros(1))                 # shape-stable bucket: zeros, averaged, written back
# one rank without ANY gradient (a batch that yields no sample triples): the collective must still match up, not hang
qs = [torch.nn.Parameter(torch.zeros(s)) for s in ((4,), (2, 3))]
if rank == 0:
    for q in qs:
        q.grad = torch.full(q.shape, 6.0)
train.allreduce_gradients(qs)
for q in qs:
    assert torch.equal(q.grad, torch.full(q.shape, 3.0)), (rank, q.grad)

# ZeRO-style path: reduce-scatter + AdamW on this rank's slice (moments sharded) + all-gather, against the replicated all-reduce path,
# bit for bit in float32 over three steps (same elementwise update injected into both; the HIP kernel is the product's)
def adamw(p, g, m, v, step, lr, betas, eps, wd):
    b1, b2 = betas
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt_().add_(eps), value=-lr)
gen = torch.Generator().manual_seed(5)
shapes = ((7,), (3, 5), (2, 2, 3), (1,))
init = [torch.randn(s, generator=gen) for s in shapes]
pa = [torch.nn.Parameter(t.clone()) for t in init]
pb = [torch.nn.Parameter(t.clone()) for t in init]
opt = train.ShardedAdamW(pb, lr=1e-2, update=adamw)
assert opt.m.numel() == (sum(t.numel() for t in init) + 1) // 2          # the moments cover half of the parameters per rank
ma, va = [torch.zeros_like(t) for t in init], [torch.zeros_like(t) for t in init]
for step in range(1, 4):
    gr = torch.Generator().manual_seed(100 * step + rank)                 # every rank its own gradients
    for i, (a, b) in enumerate(zip(pa, pb)):
        g = torch.randn(a.shape, generator=gr)
        if i == 3 and rank == 1:
            a.grad = b.grad = None                                         # ... and one of them missing on one rank
        else:
            a.grad, b.grad = g.clone(), g.clone()
    train.allreduce_gradients(pa)
    for a, m, v in zip(pa, ma, va):
        adamw(a.data, a.grad, m, v, step, 1e-2, (0.9, 0.999), 1e-8, 0.01)
    opt.step()
    for a, b in zip(pa, pb):
        assert torch.equal(a.data, b.data), (rank, step, (a.data - b.data).abs().max())
# ZeRO-3 ownership (partition_params): a rank keeps only its slice of the masters; the full parameters exist only after gather().  Same gradients,
# same results as the all-reduce path, bit for bit -- and with the released buffer poisoned, a forward without gather() would see NaN
pc = [torch.nn.Parameter(t.clone()) for t in init]
pd_ = [torch.nn.Parameter(t.clone()) for t in init]
optz = train.ShardedAdamW(pd_, lr=1e-2, update=adamw, partition_params=True, poison_released=True)
assert optz.master.numel() == optz.m.numel() == (sum(t.numel() for t in init) + 1) // 2
mc, vc = [torch.zeros_like(t) for t in init], [torch.zeros_like(t) for t in init]
for step in range(1, 4):
    optz.gather()
    for c, d in zip(pc, pd_):
        assert torch.equal(c.data, d.data), (rank, step)                   # what the forward of this step reads
    gr = torch.Generator().manual_seed(300 * step + rank)
    for c, d in zip(pc, pd_):
        g = torch.randn(c.shape, generator=gr)
        c.grad, d.grad = g.clone(), g.clone()
    train.allreduce_gradients(pc)
    for c, m, v in zip(pc, mc, vc):
        adamw(c.data, c.grad, m, v, step, 1e-2, (0.9, 0.999), 1e-8, 0.01)
    optz.step()
    assert all(torch.isnan(d.data).all() for d in pd_)                     # released: nothing but the owned slices survives the update
optz.gather(); optz.gather()                                               # (idempotent)
for c, d in zip(pc, pd_):
    assert torch.equal(c.data, d.data), (rank, "final")
# with clipping the two paths differ only by the rounding of the norm
for a, b in zip(pa, pb):
    a.grad, b.grad = torch.ones_like(a) * (rank + 1), torch.ones_like(b) * (rank + 1)
n = opt.step(max_grad_norm=1.0)
assert abs(n - 1.5 * sum(t.numel() for t in init) ** 0.5) < 1e-4, n
dist.barrier()
dist.destroy_process_group()
open(os.path.join({out!r}, "grad%d.ok" % rank), "w").write("ok")
"""


def test_gradient_exchange_world_size_2_gloo(tmp_path):
    """The training step's collectives on gloo with two ranks: the flattened all-reduce of replicated parameters (shape-stable when a rank
    has no gradient at all), the reduce-scatter + sharded AdamW + all-gather path, and its ZeRO-3 form (partition_params: masters owned by slice,
    all-gather in front of the forward) -- each must reproduce the all-reduce path bit for bit."""
    script = tmp_path / "gworker.py"
    script.write_text(_GRAD_WORKER.format(root=ROOT, out=str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "grad0.ok").exists() and (tmp_path / "grad1.ok").exists()


def test_bench_self_launches_n_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE around it must spawn its own ranks (the driver starts it that way) and
    rank 0 must print the JSON line with n_gpus=2; --launch-check stops after the rendezvous so it runs without a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=240, env=dict(env, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import json
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines == [{"launch_check": True, "n_gpus": 2, "local_rank": 0}], r.stdout[-2000:]
    # already wrapped by a launcher whose world disagrees with the flag: refuse, do not spawn again
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "disagree" in r.stderr


def test_pixel_latent_vector_export_matches_reference_loop(tmp_path):
    """ldiffusion_amd.pixel_latent_vector (vectorised) writes byte for byte the CSV of the reference's per-pixel dict loop
    (pixel_latent_vector.py:85-101, restated with the csv module in oracle/pixel_export.py)."""
    import numpy as np
    from ldiffusion_amd import pixel_latent_vector as plv
    from oracle import pixel_export as ope
    rng = np.random.default_rng(0)
    for (n, h, w) in [(5, 7, 9), (1, 3, 4), (20, 16, 12)]:
        feats = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
        label = rng.integers(0, 11, size=(h, w)).astype(np.uint8)
        assert plv.generate_title(n) == ope.generate_title(n)
        tab = plv.pixel_table(torch.from_numpy(feats), torch.from_numpy(label))
        assert tab.shape == (h * w, n + 1) and tab[w + 2, :n].tolist() == feats[:, 1, 2].tolist() and tab[w + 2, n] == label[1, 2]
        p = tmp_path / f"pixel_dict_{n}.csv"
        plv.write_pixel_csv(str(p), torch.from_numpy(feats), torch.from_numpy(label))
        assert p.read_bytes() == ope.pixel_csv_bytes([feats[k] for k in range(n)], label)
    with pytest.raises(ValueError):
        plv.pixel_table(np.zeros((2, 3, 3), np.uint8), np.zeros((4, 3), np.uint8))


_STEP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from ldiffusion_amd import train
dist.init_process_group("gloo")
rank = dist.get_rank()
# The per-batch dispatch of train_ldiffusion (train.run_step) with the two device steps replaced by stand-ins that do what the real ones do
# AFTER the backward pass: finish_step = gradient all-reduce -> norm -> AdamW.  Rank 1's batches are all background (no sample triples):
# it must still enter every collective (ADVICE round 3: a skipped step paired rank 0's gradient bucket with the epoch-end reduction).
ps = [torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2, 2))]
calls = []
def upd(params, grads, state, lr, weight_decay):
    state["step"] = state.get("step", 0) + 1
    for p, g in zip(params, grads):
        p.data -= lr * g
def fake(kind):
    def f(*a, **kw):
        pairs, state = (a[4], a[5]) if kind == "graph" else (a[8], a[9])
        n = sum(len(t) for t in pairs)
        for p in ps:
            p.grad = torch.full_like(p, float(n))           # zero gradients on the rank without triples
        train.finish_step(ps, state, 0.5, 0.0, None, update=upd)
        calls.append(kind)
        return float(n)
    return f
train.train_step, train.train_step_graphed = fake("eager"), fake("graph")
class G:   # the attributes run_step reads from a captured GraphedStep
    hidden = torch.zeros(2, 6, 8); timesteps = [801, 1]; bi = torch.zeros(64); noisy = [torch.zeros(2, 4, 8, 8)]
z0, text = torch.zeros(2, 4, 8, 8), torch.zeros(2, 6, 8)
state = {{}}
for it in range(3):
    pairs = [[(0, 1, [2, 3])], [(4, 5, [6, 7])]] if rank == 0 else [[], []]
    val, kind = train.run_step(G, None, None, None, z0, text, [801, 1], None, pairs, state, lr=1e-5)
    assert kind == ("graph" if rank == 0 else "eager"), (rank, kind)
t = torch.tensor([float(rank)], dtype=torch.float64)        # the epoch-end reduction (_reduce_mean) must pair with itself
dist.all_reduce(t)
assert t.item() == 1.0
assert len(calls) == 3 and state["step"] == 3
for p in ps:                                                # averaged gradient (2 + 0) / 2 = 1 per step, lr 0.5, three steps
    assert torch.allclose(p.data, torch.full_like(p, 1.0 - 1.5)), (rank, p.data)
dist.barrier()
dist.destroy_process_group()
open(os.path.join({out!r}, "step%d.ok" % rank), "w").write("ok")
"""


def test_no_rank_skips_the_step_world_size_2_gloo(tmp_path):
    """One rank with an all-background label (no sample triples) and one with triples: both enter the same gradient collective every batch."""
    script = tmp_path / "sworker.py"
    script.write_text(_STEP_WORKER.format(root=ROOT, out=str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / "step0.ok").exists() and (tmp_path / "step1.ok").exists()


def test_non_finite_gradient_skips_the_update_and_halves_the_loss_scale():
    """an inf / NaN gradient behind the float16 loss scale must not reach AdamW -- the step is skipped with the
    parameters, moments and step count untouched, and the loss scale halves.  Both optimizer paths (replicated, ShardedAdamW)."""
    from ldiffusion_amd import train
    ran = []

    def upd(params, grads, state, lr, weight_decay):
        state["step"] = state.get("step", 0) + 1
        ran.append(state["step"])

    for bad in (float("inf"), float("nan")):
        ps = [torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(2))]
        ps[0].grad, ps[1].grad = torch.tensor([1.0, bad, 0.0, 2.0]), torch.ones(2)
        state = {"loss_scale": 1024.0}
        assert train.finish_step(ps, state, 1e-3, 0.0, 1.0, update=upd) is False
        assert state["loss_scale"] == 512.0 and state["skipped_steps"] == 1 and "step" not in state
        assert torch.equal(ps[1].grad, torch.ones(2))                     # not scaled by inf * 0
    ps[0].grad = torch.tensor([3.0, 0.0, 0.0, 0.0]); ps[1].grad = torch.tensor([4.0, 0.0])
    assert train.finish_step(ps, state, 1e-3, 0.0, 1.0, update=upd) is True and ran == [1]
    assert torch.allclose(ps[0].grad, torch.tensor([0.6, 0, 0, 0])) and state["loss_scale"] == 512.0   # clipped to norm 1, scale kept

    def adamw(p, g, m, v, step, lr, betas, eps, wd):
        p.sub_(lr * g)

    qs = [torch.nn.Parameter(torch.ones(5))]
    opt = train.ShardedAdamW(qs, lr=0.1, update=adamw)
    qs[0].grad = torch.tensor([1.0, float("inf"), 0, 0, 0])
    state = {}
    assert train.finish_step(qs, state, 0.1, 0.0, 1.0, optimizer=opt) is False
    assert opt.step_count == 0 and opt.skipped == 1 and torch.equal(qs[0].data, torch.ones(5)) and state["loss_scale"] == train.LOSS_SCALE / 2
    qs[0].grad = torch.tensor([2.0, 0, 0, 0, 0])
    assert train.finish_step(qs, state, 0.1, 0.0, None, optimizer=opt) is True
    assert opt.step_count == 1 and torch.allclose(qs[0].data, torch.tensor([0.8, 1, 1, 1, 1]))


def test_ctypes_struct_mirrors_have_the_headers_layout(tmp_path):
    """Every struct the C ABI passes by pointer (include/ldiff.h) against its ctypes mirror in _lib.py: same size and the same offset for every
    field, as gcc lays the header out -- an appended or reordered field on one side only would otherwise be read as garbage, not fail."""
    pairs = {"ldiff_unet_cfg": _lib.UNetCfg, "ldiff_vae_cfg": _lib.VaeCfg, "ldiff_conv_args": _lib.ConvArgs, "ldiff_prof_row": _lib.ProfRow}
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ldiff.h")).read(), flags=re.S)
    assert set(re.findall(r"typedef struct\s*{[^}]*}\s*(ldiff_[a-z0-9_]+)\s*;", hdr)) == set(pairs), "a struct of the header has no ctypes mirror in this test"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ldiff.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} . %zu\\n", sizeof({cname}));')
        lines += [f'  printf("{cname} {f[0]} %zu\\n", offsetof({cname}, {f[0]}));' for f in cls._fields_]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines + ["  return 0;", "}"]))
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "layout")], check=True)
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in subprocess.run([str(tmp_path / "layout")], check=True, capture_output=True, text=True).stdout.splitlines()}
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct\s*{([^}]*)}\s*" + cname, hdr).group(1)
        names = [n for decl in body.split(";") for n in re.findall(r"(\w+)\s*(?:\[[^\]]*\])?\s*(?:,|$)", re.sub(r"^\s*(?:const\s+)?\w+\s*\*?", "", decl.strip()))]
        assert names == [f[0] for f in cls._fields_], f"{cname}: field order differs: {names} vs {[f[0] for f in cls._fields_]}"
        assert got[(cname, ".")] == C.sizeof(cls), f"{cname}: sizeof {got[(cname, '.')]} vs ctypes {C.sizeof(cls)}"
        for f in cls._fields_:
            assert got[(cname, f[0])] == getattr(cls, f[0]).offset, f"{cname}.{f[0]}: offset {got[(cname, f[0])]} vs ctypes {getattr(cls, f[0]).offset}"
