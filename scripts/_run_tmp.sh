mkdir -p gpurun_out/r4y; O=gpurun_out/r4y; rm -f $O/*
for r in 2 0 3; do echo "== LDIFF_C3D_RUN=$r" >> $O/s.txt; LDIFF_C3D_RUN=$r LDIFF_CONV3X3_DATAFLOW=2 timeout 600 python scripts/stress_c3d.py 150 2>&1 | grep -v amdgpu >> $O/s.txt; done
cat $O/s.txt
