#!/bin/bash
# MFMA-busy / wait counters of the dataflow conv kernel alone on the chip (round-4 verdict item 3a).  rocprofv3 --pmc passes over scripts/bench_conv.py
# (program straight after `--`), persistent (LDIFF_C3D_RUN=0) and one-unit runs (LDIFF_C3D_RUN=1); summary only is kept.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_pmc_c3d; rm -rf "$O"; mkdir -p "$O"
for run in 0 1; do
  export LDIFF_C3D_RUN=$run
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/run${run}_p$i" -- python3 scripts/bench_conv.py vae512_128_128_gn vae256_256_256_gn --iters 3 > "$O/run${run}_p$i.log" 2>&1 || echo "pass failed: $(tail -2 $O/run${run}_p$i.log)"
  done
  echo "== LDIFF_C3D_RUN=$run (0 = persistent, 1 = one unit per workgroup)" >> "$O/summary.txt"
  python3 scripts/pmc_summary2.py "$O" conv3x3d >> "$O/summary.txt" 2>&1
  grep -h "vae" "$O"/run${run}_p1.log >> "$O/summary.txt"
  find "$O" -name "*.csv" -delete; find "$O" -name "*.db" -delete
done
cat "$O/summary.txt"
