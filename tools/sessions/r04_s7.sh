#!/bin/bash
# round 4, GPU session 7: suite on the new head / GEMM build, dictionary-training + ISTA profiles, step API trace, bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s7
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q --maxfail=15 > "$OUT/pytest_gpu.txt" 2>&1
tail -5 "$OUT/pytest_gpu.txt"
cd /tmp && export TMPDIR=/tmp
stats() {
    local name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o stats -- python3 "$@" \
        > "$OUT/${name}_under_rocprof.txt" 2> "$OUT/$name.err"
    cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv" 2>/dev/null
    find "$OUT/$name" -name '*kernel_trace.csv' -size +8M -delete
    find "$OUT/$name" -name "*.db" -delete
}
python3 "$ROOT/tools/snmf_profile.py" 32768 513 1000 20 > "$OUT/snmf_unprofiled.txt" 2>&1
python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 25 > "$OUT/ista_unprofiled.txt" 2>&1
stats snmf_train "$ROOT/tools/snmf_profile.py" 32768 513 1000 20
stats ista "$ROOT/tools/ista_profile.py" 32768 513 2000 25
python3 "$ROOT/tools/step_api_trace.py" 32 500 257 100 5 20 > "$OUT/step_api_unprofiled.txt" 2>&1
timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d "$OUT/step_api" -o api -- python3 "$ROOT/tools/step_api_trace.py" 32 500 257 100 5 20 \
    > "$OUT/step_api_under_rocprof.txt" 2> "$OUT/step_api.err"
cp "$(find "$OUT/step_api" -name '*hip_api_stats.csv' | head -1)" "$OUT/step_hip_api_stats.csv" 2>/dev/null
find "$OUT/step_api" -name '*trace.csv' -size +8M -delete
find "$OUT/step_api" -name "*.db" -delete
cat "$OUT/snmf_unprofiled.txt" "$OUT/ista_unprofiled.txt" "$OUT/step_api_unprofiled.txt" | tail -8
python3 "$ROOT/bench.py" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
tail -c 1500 "$OUT/bench_full.json"
