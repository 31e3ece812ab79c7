#!/bin/bash
# round 4, GPU session 9: suite, dictionary training with fused objective + odd-row tail, step API delta, chain skeleton timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s9
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q --maxfail=15 > "$OUT/pytest_gpu.txt" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2
grep -E "^E  " "$OUT/pytest_gpu.txt" | head -30
bash tools/probes/run_xcd_chain_probe3.sh > "$OUT/probe3.txt" 2>&1
cat "$OUT/probe3.txt"
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
python3 "$ROOT/tools/step_api_trace.py" 32 500 257 100 5 40 > "$OUT/step_unprofiled.txt" 2>&1
cat "$OUT/snmf_unprofiled.txt" "$OUT/step_hip_api_delta.txt" "$OUT/step_unprofiled.txt" | tail -30
find "$OUT" -name '*trace.csv' -size +4M -delete; find "$OUT" -name "*.db" -delete
