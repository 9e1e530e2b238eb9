mkdir -p gpurun_out/r5h; O=gpurun_out/r5h; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -s -k "sharded" 2>&1 | grep -E "graphed steps|worst|passed|failed|Error|assert" | tail -8 > $O/t.txt
cat $O/t.txt
