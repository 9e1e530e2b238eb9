mkdir -p gpurun_out/r5c; O=gpurun_out/r5c; rm -f $O/*
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -2 > $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu > $O/smoke.txt
timeout 600 python scripts/unet_launches.py > $O/unet_launches.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
bash scripts/final_profiles.sh > $O/final.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench2.json 2> $O/bench2.err
tail -2 $O/gpu_tests.txt; tail -2 $O/smoke.txt; head -1 $O/unet_launches.txt
for f in bench bench2; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['serial']['frac'],3), d['unet_step']['ms'], d['roofline'].get('traffic'))"; done
