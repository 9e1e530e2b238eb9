mkdir -p gpurun_out/r4i; O=gpurun_out/r4i; rm -f $O/*.txt
timeout 900 python scripts/debug_graph.py tiny 1 2 --poison 2>&1 | grep -v amdgpu | grep -E "replay|Error" | cut -c1-200 >> $O/t1.txt
timeout 900 python scripts/debug_graph.py tiny 3 2 --keep-eager --check 2>&1 | grep -v amdgpu | grep -E "   loss|Error" | cut -c1-120 >> $O/t1.txt
timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -x -s -k "graphed or contrastive" 2>&1 | grep -E "graphed|AdamW|passed|failed" | tail -6 >> $O/t1.txt
cat $O/t1.txt
