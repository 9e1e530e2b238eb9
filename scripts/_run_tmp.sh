mkdir -p gpurun_out/r4q; O=gpurun_out/r4q; rm -f $O/*
timeout 900 python -m pytest tests/test_gpu_autograd.py -q -m gpu -x 2>&1 | tail -2 > $O/t.txt
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x 2>&1 | tail -2 >> $O/t.txt
timeout 600 python scripts/bench_train.py --graph 2>&1 | tail -1 >> $O/t.txt
timeout 600 python scripts/bench_train.py 2>&1 | tail -1 >> $O/t.txt
timeout 600 python scripts/train_graph_prof.py 2>&1 | grep -v amdgpu >> $O/t.txt
cat $O/t.txt
