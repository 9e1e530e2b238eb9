mkdir -p gpurun_out/r4v; O=gpurun_out/r4v; rm -f $O/*
timeout 3000 python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3 > $O/gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu > $O/smoke.txt
timeout 600 python scripts/unet_launches.py > $O/unet_launches.txt 2>&1
timeout 600 python scripts/vae_launches.py > $O/vae_launches.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
bash scripts/final_profiles.sh > $O/final.log 2>&1
tail -2 $O/gpu_tests.txt; tail -3 $O/smoke.txt; head -3 $O/unet_launches.txt; tail -c 600 $O/bench.json; tail -5 $O/final.log | cut -c1-300
