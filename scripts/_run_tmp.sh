mkdir -p gpurun_out/r3s; O=gpurun_out/r3s
for rep in 1 2; do
for v in product c3d_v3; do
  if [ $v = product ]; then L=ldiffusion_amd/libldiff_hip.so; else L=build/$v/libldiff_hip.so; fi
  timeout 400 python scripts/bench_variant.py $L --steps 6 --warmup 2 --no-cpu-baseline > $O/${v}_$rep.json 2> $O/${v}_$rep.err
  python3 -c "
import json,sys
d=json.loads(open('$O/${v}_$rep.json').read().strip().splitlines()[-1])
print('$v', round(d['value'],2), round(d['ms_per_step'],2), d['roofline']['frac'], d['roofline']['serial']['frac'], d['roofline']['serial']['avg_launch_us'])
for k in d['kernels'][:6]: print('   ', k['name'], k['launches'], k['ms'])
"
done; done
