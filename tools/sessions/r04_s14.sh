#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s14
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q --maxfail=10 > "$OUT/pytest_gpu.txt" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2; grep -E "^E  " "$OUT/pytest_gpu.txt" | head -20
timeout 900 python3 tools/batch_sweep.py 400 160 224 250 512 640 896 > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
