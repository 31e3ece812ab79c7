#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s16
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q --maxfail=10 > "$OUT/pytest_gpu.txt" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2; grep -E "^E  " "$OUT/pytest_gpu.txt" | head -20
