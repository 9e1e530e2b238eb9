mkdir -p gpurun_out/r5f; O=gpurun_out/r5f; rm -f $O/*
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -2 > $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu > $O/smoke.txt
timeout 600 python scripts/unet_launches.py > $O/unet_launches.txt 2>&1
bash scripts/final_profiles.sh > $O/final.log 2>&1
mkdir -p profiles_tmp; H=$(python3 -c "import bench; print(bench.kernel_source_hash())"); cp gpurun_out/final/pmc_traffic_$H.json profiles/
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -2 $O/gpu_tests.txt; tail -1 $O/smoke.txt; head -1 $O/unet_launches.txt
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['serial']['frac'],3), d['unet_step']['ms'], d['roofline'].get('traffic'), d['cpu_baseline']['value'])"
