#!/bin/bash
# SQ counters of the frame-parallel ISTA products in both matrix modes (separate pass, kernel-trace only)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-x3pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16x3; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq_$mode" -o p -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 4 $mode > /dev/null 2> "$OUT/pmc_sq_$mode.err"
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA \
      -d "$OUT/pmc_lds_$mode" -o p -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 4 $mode > /dev/null 2> "$OUT/pmc_lds_$mode.err"
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_fetch_$mode" -o p -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 4 $mode > /dev/null 2> "$OUT/pmc_fetch_$mode.err"
  python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_${mode}_summary.json" "$OUT/pmc_sq_$mode" "$OUT/pmc_lds_$mode" "$OUT/pmc_fetch_$mode"
done
find "$OUT" -maxdepth 1 -type d -name 'pmc_*' -exec rm -rf {} +
cat "$OUT"/pmc_*_summary.json | head -150
