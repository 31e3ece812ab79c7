#!/bin/bash
# round 4, GPU session 8: interleaved split enqueue, parallel W update, full-size test diagnostics, 8 ranks on one GPU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s8
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q --maxfail=15 > "$OUT/pytest_gpu.txt" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
grep -E "^E  " "$OUT/pytest_gpu.txt" | head -30
{
timeout 900 python3 tools/batch_sweep.py 2000 250
timeout 900 python3 tools/batch_sweep.py 400 64 128 250 512 1024 2048
DRNMF_SPLIT=1 timeout 900 python3 tools/batch_sweep.py 2000 250
} > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/snmf_profile.py" 32768 513 1000 20 > "$OUT/snmf_unprofiled.txt" 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/snmf_train" -o stats -- python3 "$ROOT/tools/snmf_profile.py" 32768 513 1000 20 \
    > "$OUT/snmf_train_under_rocprof.txt" 2> "$OUT/snmf_train.err"
cp "$(find "$OUT/snmf_train" -name '*kernel_stats.csv' | head -1)" "$OUT/snmf_train_kernel_stats.csv" 2>/dev/null
for n in 0 20; do
  timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d "$OUT/step_api_$n" -o api -- python3 "$ROOT/tools/step_api_trace.py" 32 500 257 100 5 $n \
      > "$OUT/step_api_${n}.txt" 2> "$OUT/step_api_$n.err"
  cp "$(find "$OUT/step_api_$n" -name '*hip_api_stats.csv' | head -1)" "$OUT/step_hip_api_stats_$n.csv" 2>/dev/null
done
python3 "$ROOT/tools/hip_api_delta.py" "$OUT/step_hip_api_stats_0.csv" "$OUT/step_hip_api_stats_20.csv" 20 > "$OUT/step_hip_api_delta.txt" 2>&1
cat "$OUT/snmf_unprofiled.txt" "$OUT/step_hip_api_delta.txt" | tail -30
find "$OUT" -name '*trace.csv' -size +4M -delete; find "$OUT" -name "*.db" -delete
cd "$ROOT"
DRNMF_BENCH_BACKEND=gloo DRNMF_BENCH_DEVICE=0 DRNMF_DP_BACKEND=torch timeout 900 python3 bench.py --gpus 8 --batch 4 --frames 8 --steps 2 --warmup 1 \
    --no-extras --no-cpu-baseline > "$OUT/n8_gloo_one_gpu.json" 2> "$OUT/n8_gloo_one_gpu.err"
echo "n8 rc=$?"; tail -c 600 "$OUT/n8_gloo_one_gpu.json"; tail -3 "$OUT/n8_gloo_one_gpu.err"
