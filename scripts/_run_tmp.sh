mkdir -p gpurun_out/r4u; O=gpurun_out/r4u; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -2 > $O/t.txt
for m in 0 -1 0 -1; do LDIFF_GEMM_MFAST=$m timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/b.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('mfast $m', round(d['value'],2), round(d['ms_per_step'],2), round(d['unet_step']['ms'],2))" >> $O/t.txt; done
cat $O/t.txt
