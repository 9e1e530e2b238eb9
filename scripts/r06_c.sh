#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06d; rm -rf "$O"; mkdir -p "$O"
timeout 1500 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_train.py -m gpu -q -s > "$O/pytest_train.log" 2>&1; tail -4 "$O/pytest_train.log"
timeout 600 python3 scripts/decoder_headroom.py > "$O/decoder_headroom.txt" 2>&1; tail -6 "$O/decoder_headroom.txt"
timeout 900 python3 scripts/stress_sampler.py > "$O/stress_sampler.txt" 2>&1; tail -3 "$O/stress_sampler.txt"
