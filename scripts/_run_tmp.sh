mkdir -p gpurun_out/r3v; O=gpurun_out/r3v
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -15 > $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 600 python scripts/bench_train.py > $O/bench_train.txt 2>&1
tail -5 $O/gpu_tests.txt; tail -6 $O/smoke.txt; tail -8 $O/bench_train.txt
