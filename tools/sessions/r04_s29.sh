#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s29
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 1200 python3 -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > "$OUT/tests.txt"
cat "$OUT/tests.txt"
{
timeout 300 python3 tools/train_profile.py 32 500 257 100 5 20 2>&1 | tail -1
timeout 300 python3 tools/train_profile.py 32 500 257 1000 5 10 2>&1 | tail -1
} > "$OUT/ab.txt" 2>&1
cut -c1-900 "$OUT/ab.txt"
