mkdir -p gpurun_out/r5b; O=gpurun_out/r5b; rm -f $O/*
for m in 0 1 0 1; do echo "== LDIFF_ATTN_XCD=$m" >> $O/ab.txt; LDIFF_ATTN_XCD=$m timeout 200 python scripts/bench_conv.py attn_ --iters 30 2>&1 | grep -E "attn_" >> $O/ab.txt; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "attn or attention" 2>&1 | tail -2 >> $O/ab.txt
cat $O/ab.txt
