mkdir -p gpurun_out/r4w; O=gpurun_out/r4w; rm -f $O/*
for m in 0 1 2 3; do echo "== LDIFF_GEMM_TILE=$m" >> $O/ab.txt; LDIFF_GEMM_TILE=$m timeout 200 python scripts/bench_conv.py lin_ --iters 30 2>&1 | grep -E "lin_" >> $O/ab.txt; done
python3 - <<'PY'
import re,collections
d=collections.defaultdict(dict); cur=None
for l in open('gpurun_out/r4w/ab.txt'):
    if l.startswith('=='): cur=l.split('=')[-1].strip(); continue
    m=re.match(r"(\S+)\s+([\d.]+) us",l)
    if m: d[m.group(1)][cur]=float(m.group(2))
for k,v in d.items(): print(f"{k:26s}", "  ".join(f"{t}:{v.get(t,0):7.1f}" for t in "0123"))
PY
