#!/bin/bash
# round 4, GPU session 12: split counts with two row blocks in the sub-batches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s12
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
{
timeout 900 python3 tools/batch_sweep.py 400 64 80 96 128 192 250 320 384 512 768 1024 2048
for s in 2 3 4; do DRNMF_SPLIT=$s timeout 900 python3 tools/batch_sweep.py 400 192 250 320 384 512 768; done
for s in 3 5 6; do DRNMF_SPLIT=$s timeout 900 python3 tools/batch_sweep.py 400 1024; done
DRNMF_SPLIT=1 timeout 900 python3 tools/batch_sweep.py 400 80 96
DRNMF_SPLIT=2 timeout 900 python3 tools/batch_sweep.py 400 80
} > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"
timeout 1200 python3 -m pytest tests -m gpu -q -x -k "parity or fullsize" 2>&1 | tail -3
