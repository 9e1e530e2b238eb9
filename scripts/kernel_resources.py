"""Print VGPR/AGPR/spill/occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: print(cur)
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:70]}
    else:
        cur[k.split(" [")[0]] = v
if cur: print(cur)
