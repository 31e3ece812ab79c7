#!/bin/bash
# A/B of the frame-parallel ISTA: matrix modes x XCD-aware tile map (unprofiled, then kernel-trace stats).
# DRNMF_NT_XCD is a measurement aid: build with DRNMF_MEASURE=1 for the identity-order arm to take effect.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-x3ab}
mkdir -p "$OUT"
for xcd in 1 0; do for mode in f32 bf16x3; do
  echo "== DRNMF_NT_XCD=$xcd $mode"; DRNMF_NT_XCD=$xcd python3 tools/ista_profile.py 32768 513 2000 25 $mode 2>&1 | tail -1
  DRNMF_NT_XCD=$xcd python3 tools/ista_profile.py 32768 257 2000 25 $mode 2>&1 | tail -1
done; done
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16x3; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ista_$mode" -o stats -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 25 $mode > "$OUT/ista_$mode.txt" 2> "$OUT/ista_$mode.err"
    cp "$(find "$OUT/ista_$mode" -name '*kernel_stats.csv' | head -1)" "$OUT/ista_${mode}_kernel_stats.csv" 2>/dev/null
    head -3 "$OUT/ista_${mode}_kernel_stats.csv" | cut -c1-200
    grep gemm_nt "$OUT/ista_$mode/stats_kernel_trace.csv" | awk -F'","' '{print $12, $13, $14}' | sort | uniq -c
    find "$OUT/ista_$mode" -name '*kernel_trace.csv' -delete; find "$OUT/ista_$mode" -name "*.db" -delete
done
