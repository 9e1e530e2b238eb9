mkdir -p gpurun_out/r4c; O=gpurun_out/r4c; rm -f $O/*.txt
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -s -k "graphed or fixture" 2>&1 | grep -E "graphed step|infonce fixture|passed|failed|Error|error|Segm|Warning" | tail -20 > $O/t1.txt
timeout 600 python scripts/bench_train.py --graph 2>&1 | tail -2 >> $O/t1.txt
cat $O/t1.txt
