#!/bin/bash
# round 4, GPU session 6: full GPU suite (no -x)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s6
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q -x --maxfail=15 > "$OUT/pytest_gpu.txt" 2>&1
tail -40 "$OUT/pytest_gpu.txt"
