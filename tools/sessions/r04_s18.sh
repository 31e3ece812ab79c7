#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s18
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
{
DRNMF_RB=2 timeout 600 python3 tools/batch_sweep.py 400 80 128 192
DRNMF_RB=2 DRNMF_SPLIT=4 timeout 600 python3 tools/batch_sweep.py 400 250
DRNMF_RB=2 DRNMF_SPLIT=3 timeout 600 python3 tools/batch_sweep.py 400 250
DRNMF_RB=2 DRNMF_SPLIT=8 timeout 600 python3 tools/batch_sweep.py 400 512 1024
DRNMF_KS=1 timeout 600 python3 tools/batch_sweep.py 400 250
DRNMF_KS=4 timeout 600 python3 tools/batch_sweep.py 400 250
} > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"
