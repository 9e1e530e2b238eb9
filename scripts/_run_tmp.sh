mkdir -p gpurun_out/r4n; O=gpurun_out/r4n; rm -f $O/*
for r in 2 0; do LDIFF_C3D_RUN=$r timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dataflow" 2>&1 | tail -1 >> $O/test.txt; done
cat $O/test.txt
bash scripts/ab_c3d.sh r4n c3d_v4 | grep -E "==|vae512|vae256|vae128"
for v in ldiffusion_amd build/c3d_v4; do timeout 400 python scripts/bench_variant.py $v/libldiff_hip.so --steps 6 --warmup 2 --no-cpu-baseline > $O/b.json 2>/dev/null; python3 -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('$v', round(d['value'],2), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['serial']['frac'],3), round(d['roofline']['serial']['avg_launch_us'],1))"; done
