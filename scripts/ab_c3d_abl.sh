# same-box timing of ablation builds of the dataflow conv (diagnostic; results of the ablated builds are wrong by construction):
# scripts/ab_c3d_abl.sh <outdir> <variant dirs under build/ ...>; LDIFF_C3D_RUN=0 (one persistent workgroup per CU), plain layers
O=gpurun_out/$1; shift; mkdir -p $O
export LDIFF_C3D_RUN=0
for v in product "$@" product; do
  if [ $v = product ]; then unset LDIFF_LIB; else export LDIFF_LIB=build/$v/libldiff_hip.so; fi
  echo "== $v" >> $O/ab.txt
  timeout 120 python scripts/bench_conv.py vae --iters 20 2>&1 | grep -E "_gn" | grep -v "128_3" >> $O/ab.txt
done
cat $O/ab.txt
