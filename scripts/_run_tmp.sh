mkdir -p gpurun_out/r4z; O=gpurun_out/r4z; rm -f $O/*
timeout 600 python scripts/bench_modes.py 2>&1 | grep -v amdgpu > $O/modes.txt
for i in 1 2 3; do timeout 400 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench run', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['serial']['frac'],3), d['roofline'].get('traffic'))" >> $O/modes.txt; done
bash scripts/final_profiles.sh > $O/final.log 2>&1
cat $O/modes.txt; tail -3 $O/final.log | cut -c1-200
