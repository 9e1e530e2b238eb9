mkdir -p gpurun_out/r3o
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dataflow" 2>&1 | tail -15 > gpurun_out/r3o/test.txt; cat gpurun_out/r3o/test.txt
LDIFF_C3D_RUN=0 bash scripts/ab_c3d.sh r3o abl_nostore c3d_v3 | grep -E "==|vae512|vae256|vae128"
