bash scripts/pmc_conv.sh vae512_128_128_gn r4x_plain > /dev/null 2>&1
LDIFF_BENCH_RES=1 LDIFF_BENCH_STATS=1 bash scripts/pmc_conv.sh vae512_128_128_gn r4x_res > /dev/null 2>&1
LDIFF_CONV3X3_DATAFLOW=0 bash scripts/pmc_conv.sh vae512_128_128_gn r4x_old > /dev/null 2>&1
echo "== dataflow plain"; cat gpurun_out/r4x_plain/summary.txt; echo "== dataflow res+stats"; cat gpurun_out/r4x_res/summary.txt; echo "== 8x16 plain"; cat gpurun_out/r4x_old/summary.txt
