#!/bin/bash
# round 4, GPU session 1: where do the large-batch launches go?  (run from the repo root on the GPU box)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
{
for b in 64 128 256 512 1024; do
  timeout 300 python3 "$ROOT/tools/profile_shape.py" $b 40 513 1000 25
done
DRNMF_RB=1 timeout 300 python3 "$ROOT/tools/profile_shape.py" 256 40 513 1000 25
DRNMF_RB=1 timeout 300 python3 "$ROOT/tools/profile_shape.py" 1024 40 513 1000 25
DRNMF_KS=2 timeout 300 python3 "$ROOT/tools/profile_shape.py" 256 40 513 1000 25
} > "$OUT/profile_shape.txt" 2> "$OUT/profile_shape.err"
{
timeout 600 python3 "$ROOT/tools/two_stream_probe.py" 256 200
timeout 600 python3 "$ROOT/tools/two_stream_probe.py" 512 100
timeout 600 python3 "$ROOT/tools/two_stream_probe.py" 128 200
} > "$OUT/two_stream.txt" 2> "$OUT/two_stream.err"
# kernel stats + PMC at the reference's inference slab (B = 250) and at B = 1024
for b in 250 1024; do
  ARGS="--forward-only --batch $b --frames 40 --steps 1 --warmup 1 --no-cpu-baseline --no-extras"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_b$b" -o s -- python3 "$ROOT/bench.py" $ARGS \
      > "$OUT/stats_b$b.json" 2> "$OUT/stats_b$b.err"
  cp "$(find "$OUT/stats_b$b" -name '*kernel_stats.csv' | head -1)" "$OUT/b${b}_kernel_stats.csv" 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_fetch_b$b" -o p -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch_b$b.err"
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_write_b$b" -o p -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_write_b$b.err"
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq_b$b" -o p -- python3 "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq_b$b.err"
  python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_b${b}_summary.json" "$OUT/pmc_fetch_b$b" "$OUT/pmc_write_b$b" "$OUT/pmc_sq_b$b"
done
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name "*.db" -delete
cat "$OUT/profile_shape.txt" "$OUT/two_stream.txt"
tail -5 "$OUT/profile_shape.err" "$OUT/two_stream.err"
