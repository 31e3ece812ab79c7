#!/bin/bash
# kernel-trace stats of the frame-parallel ISTA in both matrix modes:  gpurun -- 'bash tools/x3_prof.sh tag'
set -u
TAG=${1:-x3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16x3; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/ista_$mode" -o stats -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 25 $mode > "$OUT/ista_$mode.txt" 2> "$OUT/ista_$mode.err"
    cp "$(find "$OUT/ista_$mode" -name '*kernel_stats.csv' | head -1)" "$OUT/ista_${mode}_kernel_stats.csv" 2>/dev/null
    find "$OUT/ista_$mode" -name '*kernel_trace.csv' -size +8M -delete
    find "$OUT/ista_$mode" -name "*.db" -delete
    head -6 "$OUT/ista_${mode}_kernel_stats.csv" | cut -c1-200
done
