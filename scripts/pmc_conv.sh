#!/bin/bash
# PMC passes over one conv shape (diagnostic).  usage: scripts/pmc_conv.sh <shape-filter> <out-name>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
S=$1; O=gpurun_out/$2; rm -rf "$O"; mkdir -p "$O"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$O/p$i" -- python3 scripts/bench_conv.py "$S" --iters 3 > "$O/p$i.log" 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
python3 scripts/pmc_summary.py "$O" conv3x3 > "$O/summary.txt" 2>&1
cat "$O/summary.txt"
find "$O" -name "*.csv" -delete; find "$O" -name "*.db" -delete
