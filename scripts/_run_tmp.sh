mkdir -p gpurun_out/r4l; O=gpurun_out/r4l; rm -f $O/*
for rep in 1 2; do for r in 2 3 4; do
  LDIFF_C3D_RUN=$r timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/run_${r}_$rep.json 2> $O/err.txt
  python3 -c "
import json
d=json.loads(open('$O/run_${r}_$rep.json').read().strip().splitlines()[-1])
print('run $r', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['serial']['frac'],3))"
done; done
