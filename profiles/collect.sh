#!/bin/bash
# Round evidence recipe; run from the repo root on the GPU box:
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r02'
# Writes under gpurun_out/<tag>/ (scratch); copy what should be judged into profiles/.
# PMC passes are separate runs with --kernel-trace only (never with sys/hip/hsa traces).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
# 1. the default bench line
python3 "$B" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
# 2. kernel trace + stats of the headline workload alone (1 step; the extra lines of the default
#    command reuse the same kernel templates at other shapes and would blur the per-kernel means)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 "$B" --steps 1 --warmup 1 \
    --no-cpu-baseline --no-ista --no-train --no-slab --no-config5 > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
cp "$OUT"/stats/*/stats_kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || \
    cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
# 3. PMC passes, 20 frames of the headline workload (cell + head only)
SMALL="--frames 20 --steps 1 --warmup 0 --no-cpu-baseline --no-ista --no-train --no-slab --no-config5"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_fetch" -o p -- python3 "$B" $SMALL \
    > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_write" -o p -- python3 "$B" $SMALL \
    > /dev/null 2> "$OUT/pmc_write.err"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o p -- python3 "$B" $SMALL \
    > /dev/null 2> "$OUT/pmc_sq.err"
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_summary.json" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
# keep the merge-back small: the raw per-dispatch CSVs are large
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name "*.db" -delete
ls -laR "$OUT" | head -60
