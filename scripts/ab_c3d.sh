# same-box A/B of dataflow conv3x3 variants (diagnostic): scripts/ab_c3d.sh <outdir> <variant dirs under build/ ...>
O=gpurun_out/$1; shift; mkdir -p $O
for v in product "$@"; do
  if [ $v = product ]; then unset LDIFF_LIB; else export LDIFF_LIB=build/$v/libldiff_hip.so; fi
  for rs in "0 0" "1 1"; do set -- $rs
    if [ $1 = 1 ]; then export LDIFF_BENCH_RES=1 LDIFF_BENCH_STATS=1; else unset LDIFF_BENCH_RES LDIFF_BENCH_STATS; fi
    echo "== $v res/stats=$1 run=${LDIFF_C3D_RUN:-2}" >> $O/ab.txt
    timeout 120 python scripts/bench_conv.py vae --iters 20 2>&1 | grep -E "_gn" | grep -v "128_3" >> $O/ab.txt
  done
done
cat $O/ab.txt
