"""bench.py -- headline benchmark of the Laplace-diffusion sampling path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: spawns the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1] -- 8 patches of
512x512 per GPU through the 5-pass sampler (VAE encode -> 5 x [UNet -> PLMS step -> VAE decode -> uint8 -> luma])
with the SD-v1.5-sized UNet/VAE (seeded synthetic weights in the diffusers key layout: no checkpoint or network
here), followed by the mask tail (stand-in linear probe over the per-pixel latent vectors -> argmax -> uint8) and,
for N > 1, the RCCL all-gather that reassembles the masks.  Patches shard across ranks with no data-path collective
(weak scaling: 8 patches per GPU).  Inputs are resident in HBM when the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- live HIP-event measurement of the dominant kernel over the timed region
  cpu_baseline -- the CPU oracle (plain-torch fp32 restatement, oracle/) timed on the host cores (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PATCHES_PER_GPU = 8
IMG = 512
N_PASSES = 5
N_CLASSES = 6
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0      # HBM3E spec, same guide
# Algorithmic work of one UNet pass at 512x512 (SURVEY.md 8d / BASELINE.md section 2)
UNET_WEIGHT_BYTES = 1.719e9
UNET_ACT_BYTES_PER_SAMPLE = 1.014e9
UNET_FLOP_PER_SAMPLE = 797.3e9
PATCH_FLOP = 17.68e12       # algorithmic flops of one 5-pass 512x512 patch: encode + 5 x (UNet + decode), SURVEY.md 8d


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle timing (rank 0, N=1)")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-launch HIP-event profile")
    ap.add_argument("--no-unet-step", action="store_true", help="skip the separate UNet-step timing (used under rocprofv3 so that the\n                    kernel mix of the whole process equals the mix of the timed region)")
    ap.add_argument("--launch-check", action="store_true", help="rendezvous only (gloo, no GPU): every rank joins, all-reduces its rank and rank 0 prints\n                    {launch_check, n_gpus}; covers the self-launch / env contract on a CPU box")
    ap.add_argument("--no-pipeline", action="store_true", help="one batch in flight: every step joins its own decodes before the next one is enqueued")
    ap.add_argument("--tiny", action="store_true", help="reduced-width graph + 64x64 patches (plumbing check only; not a valid bench line)")
    return ap.parse_args()


def usable_cpus() -> int:
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box shows
    256 logical CPUs but grants 16 -- 256 torch threads on a 16-CPU quota run 10x slower than 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def kernel_source_hash() -> str:
    """sha256 over the HIP sources and headers of the library, first 12 hex digits: names the PMC traffic file a measurement
    belongs to, so that a profile taken on other kernel code is never quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ldiffusion_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ldiffusion_amd", "csrc", "*.h"))
                    + [os.path.join(ROOT, "include", "ldiff.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the rocprofv3 PMC passes over this same workload, taken on exactly this kernel
    source (profiles/pmc_traffic_<source hash>.json, written by scripts/final_profiles.sh: (2*FETCH_SIZE + WRITE_SIZE)*1024, see its
    _note).  None when no profile of the current sources is committed: PMC counters cannot be read from inside the process, so this
    is the offline measurement, and a stale one is refused rather than quoted."""
    try:
        with open(os.path.join(ROOT, "profiles", f"pmc_traffic_{kernel_source_hash()}.json")) as f:
            return json.load(f)["kernels"][kernel_name]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def pmc_unet(batch):
    """MFMA-busy share and memory-side bytes of a UNet pass at `batch` from the rocprofv3 PMC passes of scripts/pmc_unet_pass.sh on exactly this
    kernel source (profiles/pmc_unet_<source hash>.json); None when no such profile is committed (same rule as pmc_traffic)."""
    try:
        with open(os.path.join(ROOT, "profiles", f"pmc_unet_{kernel_source_hash()}.json")) as f:
            d = json.load(f)["batches"][str(batch)]
        return {"mfma_busy": d["mfma_busy"], "memory_side_GBps": d["memory_side_GBps"], "memory_side_bytes": d["memory_side_bytes"],
                "how": "rocprofv3 --pmc over the pass with eager launches (profiler clocks), SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles and 2 x FETCH_SIZE + WRITE_SIZE; "
                       "the bytes are counted on the memory side of the L2s: Infinity-Cache hits included"}
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(ucfg, vcfg, usd, vsd, img, n_passes):
    """The CPU oracle (oracle/, plain-torch fp32 restatement) TIMED end to end on ONE patch of the bench workload: the whole
    n_passes sampler (encode, n x [UNet, PLMS step, decode, uint8, luma]) after a small warm-up of the thread pool."""
    from oracle import pipeline as op
    threads = usable_cpus()
    torch.set_num_threads(threads)
    pipe = op.OraclePipeline(op.OracleUNet(usd, ucfg), op.OracleVAE(vsd, vcfg))
    g = torch.Generator().manual_seed(1234)
    x = torch.rand((1, 3, img, img), generator=g)
    ctx = torch.randn((1, 6, ucfg["cross_attention_dim"]), generator=g) * 0.5
    op.sample_v6(pipe, torch.rand((1, 3, 64, 64)), ctx, 3)  # thread-pool / allocator warm-up at a small size
    t0 = time.perf_counter()
    op.sample_v6(pipe, x, ctx, n_passes)
    per_patch = time.perf_counter() - t0
    return {"value": 1.0 / per_patch, "unit": "patches/sec", "cores": threads, "kind": "port", "how": "timed",
            "sample": f"1 patch {img}x{img} through the whole {n_passes}-pass sampler (oracle.pipeline.sample_v6), timed end to end once "
                      f"({per_patch:.1f} s) after a 64x64 warm-up; plain-torch fp32 restatement (oracle/), torch {torch.__version__}, "
                      f"{threads} threads -- not diffusers"}


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N fresh worker processes through
    torch.distributed.run (one rank per GPU) and return their exit code.  The parent never touches the GPU and never
    re-executes itself (an exec from a process that initialised HIP takes the box down); rank 0 of the children prints the
    one JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    # launch decision first, before anything can initialise the GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if args.launch_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)])
            dist.all_reduce(t)
            assert t.item() == world * (world - 1) / 2
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "local_rank": local_rank}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: there is no CPU fallback for the product path")
    # LDIFF_BENCH_SHARED_GPU=1 (test only, not a valid bench line): all ranks on cuda:0 over gloo, to exercise the multi-rank
    # control flow (barrier, max over ranks, mask all-gather) on a single-GPU box; the real run is one rank per GPU over RCCL.
    shared = os.environ.get("LDIFF_BENCH_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from ldiffusion_amd import _lib, configs, weights
    from ldiffusion_amd.models import AutoencoderKL, UNet2DConditionModel
    from ldiffusion_amd.parallel import gather_masks, shard_range
    from ldiffusion_amd.pipeline import LaplaceSampler, StableDiffusionImg2ImgPipeline, probe_argmax_mask

    ucfg, vcfg = (configs.TINY_UNET, configs.TINY_VAE) if args.tiny else (configs.SD15_UNET, configs.SD15_VAE)
    img = 64 if args.tiny else IMG
    # every rank generates the same 0.9 G synthetic parameters on the host: share the box's cores between the ranks instead of letting each
    # one start a full-width thread pool (8 ranks x 16 threads on a 16-CPU quota), and report the setup time the driver's clock sees
    t_setup = time.perf_counter()
    torch.set_num_threads(max(1, usable_cpus() // world))
    usd = weights.synthetic_state_dict(weights.unet_param_shapes(ucfg), 42, fp16_values=True)
    vsd = weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True)
    pipe = StableDiffusionImg2ImgPipeline(AutoencoderKL(vcfg, vsd, dev), UNet2DConditionModel(ucfg, usd, dev))
    sampler = LaplaceSampler(pipe)

    total = PATCHES_PER_GPU * world
    lo, hi = shard_range(total, rank, world)
    g = torch.Generator().manual_seed(1234)
    images_all = torch.rand((total, 3, img, img), generator=g)              # ToTensor range, pixel_latent_vector.py:29-32
    ctx = (torch.randn((1, 6, ucfg["cross_attention_dim"]), generator=torch.Generator().manual_seed(1235)) * 0.5).to(dev)
    images = images_all[lo:hi].to(dev)
    hg = torch.Generator().manual_seed(1236)
    head_w = (torch.randn((N_CLASSES, N_PASSES), generator=hg) / N_PASSES ** 0.5).to(dev)  # stand-in for the out-of-scope segmentor head
    head_b = (0.1 * torch.randn(N_CLASSES, generator=hg)).to(dev)

    last_out = {}

    def finish(out):
        last_out["out"] = out
        # mask tail: linear probe over the uint8 per-pixel latent vectors + arg-max, ONE library launch (ldiff_probe_argmax_u8): no
        # vendor-library or ATen kernel runs inside the timed steps
        mask = probe_argmax_mask(out["features"], head_w, head_b, 1.0 / 255.0)
        return gather_masks(mask, total) if world > 1 else mask

    def step():
        return finish(sampler.sample(images, ctx, N_PASSES, want_features=True, want_rgb=True))

    sampler2 = LaplaceSampler(pipe)   # a second pipeline over the SAME UNet / VAE handles: its own latent / feature buffers

    def run_steps(n):
        """n steps (batches).  Default: two batches in flight -- the side-stream join is deferred (ldiff_pipeline_set_overlap mode 2), batch
        k + 1 is enqueued before batch k is joined and finished, so the trailing decodes of batch k run beside the encode and the first UNet
        pass of batch k + 1 instead of alone (same work, same results: scripts/bench_pipelined.py, +1.6 ... +2.0 % since the persistent conv
        kernels yield after every unit).  Every batch is joined and finished inside the call.  --no-pipeline: every step joins its own decodes."""
        m = None
        if args.no_pipeline or n < 2:
            for _ in range(n):
                m = step()
            return m
        sampler.set_overlap(2); sampler2.set_overlap(2)
        prev = None
        for i in range(n):
            sp = (sampler, sampler2)[i & 1]
            out = sp.sample(images, ctx, N_PASSES, want_features=True, want_rgb=True)
            if prev is not None:
                prev[0].join()
                m = finish(prev[1])
            prev = (sp, out)
        prev[0].join()
        m = finish(prev[1])
        sampler.set_overlap(1); sampler2.set_overlap(1)
        return m

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    prof = not args.no_prof
    lib = _lib.load()
    setup_s = time.perf_counter() - t_setup
    run_steps(max(args.warmup - 1, 0))
    # Per-launch HIP events on every contraction launch cost ~6 % of a step, so the full per-kernel table comes from one
    # untimed step (the last warm-up step, or an extra one when --warmup 0), and inside the timed region only the launches
    # of the dominant kernel found there carry events (260 of ~3,000 launches per step): its average launch duration is
    # measured live over the timed region, on the launch stream.
    # That untimed step also runs with the decode side stream OFF, so its rows are per-kernel times of kernels that own the
    # chip ("serial"); the timed region runs the product default (decode of pass k beside the UNet pass k+1), where launch
    # durations are stretched by the sharing and say less about the kernel.
    rows_all = []
    if prof:
        sampler.set_overlap(0)
        lib.ldiff_prof_set_filter(None)
        lib.ldiff_prof_enable(1)
        step()
        torch.cuda.synchronize()
        lib.ldiff_prof_enable(0)
        rows_all = _lib.prof_collect()
        sampler.set_overlap(1)
        lib.ldiff_prof_set_filter(max(rows_all, key=lambda r: r["ms"])["name"].encode())
    elif args.warmup > 0:
        step()
    sync()
    if prof:
        lib.ldiff_prof_enable(1)
    t0 = time.perf_counter()
    masks = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    lib.ldiff_prof_enable(0)
    rows = _lib.prof_collect() if prof else []
    lib.ldiff_prof_set_filter(None)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert masks.shape[0] == total and masks.dtype == torch.uint8
    timed_out = {k: last_out["out"][k].clone() for k in ("features", "latents")}   # of the LAST timed step (this rank's patches); the sampler reuses its buffers
    feat_crc = None
    # result check (untimed): the masks of the LAST timed step (two batches in flight, deferred joins, persistent conv kernels on short
    # runs) against one more step with everything on one stream and one batch in flight -- same inputs, so bit for bit
    import zlib
    sampler.set_overlap(0)
    ref_masks = step()
    sampler.set_overlap(1)
    torch.cuda.synchronize()
    ref_out = last_out["out"]
    checked = bool(torch.equal(masks, ref_masks)) and all(torch.equal(timed_out[k], ref_out[k]) for k in ("features", "latents"))
    masks_crc = zlib.crc32(masks.cpu().numpy().tobytes()) & 0xFFFFFFFF
    # arithmetic regression signals (the stand-in head's masks hardly move with the kernels' rounding; these do): CRC of the uint8 per-pixel
    # latent vectors and the extreme / mean magnitude of the final latents, rank 0's patches
    feat_crc = zlib.crc32(timed_out["features"].cpu().numpy().tobytes()) & 0xFFFFFFFF
    lat_absmax = float(timed_out["latents"].abs().max().item())
    lat_absmean = float(timed_out["latents"].abs().double().mean().item())
    sampler.check_finite()    # non-finite detector of both graphs (include/ldiff.h): raises if any step of this run overflowed fp16 anywhere
    if not checked:
        raise SystemExit(f"bench: the pipelined step's masks differ from the serial step's ({int((masks != ref_masks).sum())} pixels)")

    # ---- UNet step alone (the metric's second half: UNet-step HBM GB/s vs peak), HIP events on the launch stream ----
    unet_ms = unet_eager_ms = None
    unet_small = {}
    unet_nodes = {}
    if not args.no_unet_step:
        def time_unet(lat, reps=5):
            for _ in range(3):
                pipe.unet(lat, 501, ctx)       # eager, capture, first replay
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                pipe.unet(lat, 501, ctx)   # launched on torch's current stream, the same one the events are recorded on
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        def unet_step_at(b, reps=5):
            lat = torch.randn((b, 4, img // 8, img // 8), device=dev)
            g_ms = time_unet(lat, reps)            # product default: hipGraph replay of the pass's launches
            unet_nodes[b] = pipe.unet.graph_nodes
            pipe.unet.set_graph(False)
            e_ms = time_unet(lat, reps)
            pipe.unet.set_graph(True)
            return g_ms, e_ms

        unet_ms, unet_eager_ms = unet_step_at(PATCHES_PER_GPU)
        # the regime where the metric's "UNet-step HBM GB/s vs peak" binds (B <= 2 sits on the HBM ridge) and the reference's own
        # batch (segmentor.py:96, configs[0]: B = 1)
        for b in (1, 2):
            unet_small[b] = unet_step_at(b, 10)

    # ---- the reference's own batch (one patch per call: pixel_latent_vector.py:63, segmentor.py:96), whole sampler, untimed extra: latency per patch ----
    sample_b1_ms = None
    if not args.no_unet_step and rank == 0:
        one = images[:1].contiguous()
        for _ in range(2):
            sampler.sample(one, ctx, N_PASSES, want_features=True, want_rgb=True)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(5):
            sampler.sample(one, ctx, N_PASSES, want_features=True, want_rgb=True)
        torch.cuda.synchronize()
        sample_b1_ms = (time.perf_counter() - tb) / 5 * 1e3

    if dist is not None:
        dist.barrier()   # rank 0's untimed extras above are done: nobody tears the communicator down under a rank that still runs
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    result = {
        "metric": "patches/sec (512x512, 5 denoise steps)",
        "value": total * args.steps / elapsed,
        "unit": "patches/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1]: {PATCHES_PER_GPU} patches/GPU of {img}x{img}, {N_PASSES}-pass Laplace/PLMS sampler, "
                               f"SD-v1.5-size UNet (859.5M) + VAE, seeded synthetic weights, ctx L=6; +linear-probe argmax mask"
                               + (", RCCL all-gather of masks" if world > 1 else ""),
                   "patches_per_gpu": PATCHES_PER_GPU, "image": img, "n_passes": N_PASSES, "parallelism": f"dp{world} (patch sharding)",
                   "batches_in_flight": 1 if (args.no_pipeline or args.steps < 2) else 2},
    }
    result["config"]["precision"] = ("split residual stream (fp16 hi|lo), fp16 MFMA operands, fp32 accumulate; UNet mode 1, VAE encoder "
                                     "mode 2 (the lo halves of its 3x3 conv operands travel as e4m3 through the block-scaled fp8 MFMA unless LDIFF_LO8=0: "
                                     "a 2^-11 correction term, DESIGN.md section 3), decoder mode 0 (include/ldiff.h ldiff_unet_set_precision): latents within 1e-3 of the fp32 oracle, "
                                     "uint8 features within one grey level")
    result["checked"] = checked
    result["inputs"] = ("resident in HBM when the timed region starts; uint8 features and masks stay on the device (the reference copies the decoded image "
                        "to the host once per pass, pixel_latent_vector.py:81: 25 MB in + 10.5 MB out per step = 0.4 % at PCIe rates, DESIGN.md section 7)")
    result["setup_s"] = setup_s
    result["check"] = {"masks_crc32": f"{masks_crc:08x}", "classes_present": int(masks.max().item()) + 1,
                       "features_crc32": f"{feat_crc:08x}", "latents_absmax": lat_absmax, "latents_absmean": lat_absmean, "finite": True,
                       "note": "masks_crc32 is of the STAND-IN head's masks (2 of 6 classes occur; it hardly moves with the kernels' rounding): the arithmetic signal is features_crc32 / latents_*; "
                               "finite = ldiff_pipeline_check_finite passed (no fp16 overflow in either graph during the run)",
                       "how": "masks, uint8 features and final latents of the last timed step == those of an extra untimed step on ONE stream with one batch in flight (bit for bit); "
                              "the same configuration against the fp32 CPU oracle: tests/test_gpu_models.py::test_config1_b8_bench_mode_against_oracle"}
    if dist is not None:
        result["comm"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                          "collective": "all_gather_into_tensor of uint8 masks, once per step (ldiffusion_amd/parallel.py gather_masks)"}
    if not args.tiny:
        sf = total * PATCH_FLOP / elapsed * args.steps / 1e12 / world   # algorithmic TFLOP/s per GPU over the whole timed region
        result["step"] = {"algorithmic_tflops_per_gpu": sf, "mfma_frac": sf / MFMA_PEAK_TFLOPS,
                          "note": "17.68 TFLOP per patch (SURVEY 8d) x patches / wall time of the timed region: every kernel, launch gap and the "
                                  "mask tail included; split-operand contractions execute more MFMA flops than this algorithmic count"}
    if rows:
        dom = max(rows, key=lambda r: r["ms"])          # the one kernel profiled inside the timed region
        tot_ms = sum(r["ms"] for r in rows_all)
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        result["roofline"] = {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach / MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(dom["name"]), "kernel_source_hash": kernel_source_hash(),
                              "flops_counted": "executed MFMA flops (2*M*N*K of the launch as run: parity-folded upsampling convs count 16/36 of the 9-tap flops, split-operand convs twice)",
                              "launches": dom["launches"], "avg_launch_us": 1e3 * dom["ms"] / dom["launches"],
                              "algorithmic_GBps": dom["bytes"] / (dom["ms"] * 1e-3) / 1e9,
                              "share_of_profiled_time": next(r["ms"] for r in rows_all if r["name"] == dom["name"]) / tot_ms,
                              "measured": f"HIP events on the launch stream around its {dom['launches']} launches in the timed region, where the VAE "
                                          "decode (side stream) shares the chip with the next UNet pass and, with two batches in flight, with the next "
                                          "batch's encode; its workgroups retire after one unit there ON PURPOSE (ConvParams::short_runs), handing CUs to the "
                                          "other stream: a launch's wall time inside the shared region is the sharing, not the kernel (the step got "
                                          "shorter as this number fell: DESIGN.md section 7); `serial` is the kernel alone on the chip"}
        ser = next(r for r in rows_all if r["name"] == dom["name"])
        sa = ser["flops"] / (ser["ms"] * 1e-3) / 1e12
        result["roofline"]["serial"] = {"achieved": sa, "frac": sa / MFMA_PEAK_TFLOPS, "avg_launch_us": 1e3 * ser["ms"] / ser["launches"],
                                        "launches": ser["launches"], "measured": "same kernel in the untimed step with the side stream off (kernel alone on the chip)"}
        result["kernels"] = [{"name": r["name"], "launches": r["launches"], "ms": round(r["ms"], 3),
                              "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1), "GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)}
                             for r in sorted(rows_all, key=lambda r: -r["ms"])]
        result["kernels_from"] = "one untimed step with events on every contraction / GroupNorm launch and the decode side stream off (ms per step)"
    if not args.tiny and unet_ms is not None:
        ub = UNET_WEIGHT_BYTES + PATCHES_PER_GPU * UNET_ACT_BYTES_PER_SAMPLE
        uf = PATCHES_PER_GPU * UNET_FLOP_PER_SAMPLE
        result["unet_step"] = {"ms": unet_ms, "batch": PATCHES_PER_GPU,
                               "hbm_GBps": ub / (unet_ms * 1e-3) / 1e9, "hbm_frac": ub / (unet_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "tflops": uf / (unet_ms * 1e-3) / 1e12, "mfma_frac": uf / (unet_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                               "ms_eager_launches": unet_eager_ms, "launches": unet_nodes.get(PATCHES_PER_GPU), "pmc": pmc_unet(PATCHES_PER_GPU),
                               "launch": "hipGraph replay (ms) vs the same kernels launched one by one (ms_eager_launches); launches = nodes of the captured graph"}
        for b, (g_ms, e_ms) in unet_small.items():
            bb = UNET_WEIGHT_BYTES + b * UNET_ACT_BYTES_PER_SAMPLE
            result[f"unet_step_b{b}"] = {"ms": g_ms, "batch": b, "hbm_GBps": bb / (g_ms * 1e-3) / 1e9, "hbm_frac": bb / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                         "tflops": b * UNET_FLOP_PER_SAMPLE / (g_ms * 1e-3) / 1e12,
                                         "mfma_frac": b * UNET_FLOP_PER_SAMPLE / (g_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                         "ms_eager_launches": e_ms, "launches": unet_nodes.get(b), "pmc": pmc_unet(b),
                                         "algorithmic_bytes": bb}
    if sample_b1_ms is not None:
        result["sample_b1"] = {"ms_per_patch": sample_b1_ms, "patches_per_sec": 1e3 / sample_b1_ms,
                               "what": f"one {img}x{img} patch per call through the whole {N_PASSES}-pass sampler (the reference's own batch: pixel_latent_vector.py:63), "
                                       "decodes beside the next UNet pass, joined per call; wall time per call, untimed extra (not the headline)"}
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(ucfg, vcfg, usd, vsd, img, N_PASSES)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
