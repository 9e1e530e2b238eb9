#!/bin/bash
# A/B of the persistent 16x16 conv3x3 kernel inside the whole bench step, on ONE box (boxes differ by 10-15 %): off, on, off, on
for m in 0 1 0 1; do echo "== LDIFF_CONV3X3_PINGPONG=$m"; LDIFF_CONV3X3_PINGPONG=$m timeout 300 python bench.py --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step %.2f  patches/s %.2f  unet ms %.2f' % (d['ms_per_step'], d['value'], d['unet_step']['ms']))
for k in d['kernels'][:6]: print('  %-28s n=%4d ms=%8.2f tf=%7.1f GB/s=%7.1f'%(k['name'],k['launches'],k['ms'],k['tflops'],k['GBps']))
"; done
