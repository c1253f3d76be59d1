#!/bin/bash
# Stall / issue counters of the InfoNCE kernels: separate --pmc passes, --kernel-trace only, program directly after `--`.
#   bash scripts/gpu_stall_counters.sh stall_r02 [--b3]
set -e -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/${1:-stall}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python3 "$R/profiles/infonce_stall_probe.py" $2 > "$OUT/p$i.log" 2>&1 || echo "pass $i ($set) failed"
  echo "pass $i done: $set"
done
find "$OUT" -name "*.db" -delete 2>/dev/null || true
find "$OUT" -name "*kernel_trace.csv" -delete 2>/dev/null || true
