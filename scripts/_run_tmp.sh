mkdir -p gpurun_out/r4s; O=gpurun_out/r4s; rm -f $O/*
for m in 0 -1; do LDIFF_CONV3X3_IMGFAST=$m timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/b_$m.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$O/b_$m.json').read().strip().splitlines()[-1]); print('imgfast $m', round(d['value'],2), round(d['ms_per_step'],2), round(d['unet_step']['ms'],2))"; done
for m in 0 -1; do LDIFF_CONV3X3_IMGFAST=$m timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/c_$m.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$O/c_$m.json').read().strip().splitlines()[-1]); print('imgfast $m', round(d['value'],2), round(d['ms_per_step'],2), round(d['unet_step']['ms'],2))"; done
