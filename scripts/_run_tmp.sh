mkdir -p gpurun_out/r5e; O=gpurun_out/r5e; rm -f $O/*
for rep in 1 2; do for v in ldiffusion_amd build/prev; do
  LDIFF_LIB=$v/libldiff_hip.so timeout 200 python scripts/bench_conv.py _up_ --iters 20 2>&1 | grep -E "_up_" | sed "s|^|$v |" >> $O/t.txt
  timeout 400 python scripts/bench_variant.py $v/libldiff_hip.so --steps 6 --warmup 2 --no-cpu-baseline > $O/b.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v bench', round(d['value'],2), round(d['ms_per_step'],2), round(d['unet_step']['ms'],2), [ (k['name'],k['ms']) for k in d['kernels'] if k['name'] in ('conv3x3<16x16,128>','conv3x3<8x16,128>')])" >> $O/t.txt
done; done
cat $O/t.txt
