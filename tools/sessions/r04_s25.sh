#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s25
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 900 python3 -m pytest tests -m gpu -q -x -k "snmf or ista or mu or head or gemm or train or dense" 2>&1 | tail -5 > "$OUT/tests.txt"
cat "$OUT/tests.txt"
{
for v in 0 1; do
  DRNMF_THIN=$v timeout 300 python3 tools/thin_ab.py 2>&1 | tail -1
  DRNMF_THIN=$v timeout 300 python3 tools/snmf_profile.py 32768 513 1000 20 2>&1 | grep "per iteration"
done
} > "$OUT/ab.txt" 2>&1
cat "$OUT/ab.txt"
