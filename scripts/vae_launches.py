"""One SD-v1.5 VAE encode (precision 2) and one decode (precision 0) at B=8, 512x512 with HIP events on every contraction / norm
launch (LDIFF_PROF_DUMP): launches grouped by kernel and work, sorted by total time.  Diagnostic."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("LDIFF_PROF_DUMP") is None:
    for what in ("decode", "encode"):
        env = dict(os.environ, LDIFF_PROF_DUMP="1", VAE_WHAT=what)
        out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        rows = []
        for l in out.stderr.splitlines():
            m = re.match(r"\[ldiff_prof\] (\S+)\s+([\d.]+) us\s+([\d.]+) GFLOP\s+([\d.]+) TFLOP/s\s+([\d.]+) MB\s+([\d.]+) GB/s", l)
            if m:
                rows.append((m.group(1), float(m.group(2)), float(m.group(3)), float(m.group(5))))
        if not rows:
            print(out.stderr[-2000:])
        groups = {}
        for r in rows:
            g = groups.setdefault((r[0], round(r[2], 1), round(r[3], 0)), [0, 0.0])
            g[0] += 1; g[1] += r[1]
        print(f"== {what}: {len(rows)} profiled launches, {sum(r[1] for r in rows) / 1e3:.2f} ms")
        print(f"{'kernel':28s} {'GFLOP':>9s} {'MB':>8s} {'n':>3s} {'total us':>9s} {'us each':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}")
        for (name, gf, mb), (n, us) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:45]:
            print(f"{name:28s} {gf:9.1f} {mb:8.0f} {n:3d} {us:9.1f} {us / n:8.1f} {gf * n / us * 1e3:8.0f} {mb * n / us * 1e3:7.0f}")
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from ldiffusion_amd import _lib, configs, weights
from ldiffusion_amd.models import AutoencoderKL
vcfg = configs.SD15_VAE
vae = AutoencoderKL(vcfg, weights.synthetic_state_dict(weights.vae_param_shapes(vcfg), 43, fp16_values=True), "cuda:0")
what = os.environ["VAE_WHAT"]
x = torch.rand((8, 3, 512, 512), device="cuda:0")
z = torch.randn((8, 4, 64, 64), device="cuda:0")
run = (lambda: vae.decode(z).sample) if what == "decode" else (lambda: vae.encode(x).latent_dist.mean)
for _ in range(2):
    run()
torch.cuda.synchronize()
lib = _lib.load()
lib.ldiff_prof_set_filter(None)
lib.ldiff_prof_enable(1)
run()
torch.cuda.synchronize()
lib.ldiff_prof_enable(0)
_lib.prof_collect()
