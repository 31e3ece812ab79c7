#!/bin/bash
# SQ counters of the frame-parallel ISTA products in both matrix modes (separate passes, kernel-trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-x3pmc}
MODES=${2:-"f32 bf16x3"}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in $MODES; do
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq_$mode" -o p -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 4 $mode > /dev/null 2> "$OUT/pmc_sq_$mode.err"
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA \
      -d "$OUT/pmc_lds_$mode" -o p -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 4 $mode > /dev/null 2> "$OUT/pmc_lds_$mode.err"
  python3 "$ROOT/tools/x3_pmc_report.py" "$OUT" $mode
done
find "$OUT" -name '*kernel_trace.csv' -delete
