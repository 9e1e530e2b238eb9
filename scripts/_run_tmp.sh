mkdir -p gpurun_out/r5g; O=gpurun_out/r5g; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -x -s -k "train_ldiffusion" 2>&1 | grep -E "train_ldiffusion|eager loop|passed|failed|Error|assert" | tail -8 > $O/t.txt
cat $O/t.txt
