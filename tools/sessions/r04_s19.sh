#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s19
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 2700 python3 -m pytest tests -m gpu -q --maxfail=10 > "$OUT/pytest_gpu.txt" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.txt" | tail -2; grep -E "^E  " "$OUT/pytest_gpu.txt" | head -20
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), d['roofline']['frac'], d['roofline'].get('traffic'))"
