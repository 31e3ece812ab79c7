#!/bin/bash
# round 4, GPU session 11: row blocking / atom ranges of the sub-batches, split at small batches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s11
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
{
DRNMF_RB=2 timeout 600 python3 tools/batch_sweep.py 400 250 512
DRNMF_RB=1 timeout 600 python3 tools/batch_sweep.py 400 250 512 1024
DRNMF_KS=2 timeout 600 python3 tools/batch_sweep.py 400 250
DRNMF_SPLIT=2 timeout 600 python3 tools/batch_sweep.py 400 64 96
DRNMF_SPLIT=1 timeout 600 python3 tools/batch_sweep.py 400 96 128
DRNMF_SPLIT=3 timeout 600 python3 tools/batch_sweep.py 400 250 384
DRNMF_SPLIT=5 timeout 600 python3 tools/batch_sweep.py 400 1024
DRNMF_SPLIT=6 timeout 600 python3 tools/batch_sweep.py 400 1024 2048
} > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"
