#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s20
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
{
for ks in 1 2 4; do DRNMF_SPLIT=2 DRNMF_KS=$ks timeout 600 python3 tools/batch_sweep.py 400 64 48; done
for ks in 1 2 4; do DRNMF_SPLIT=1 DRNMF_KS=$ks timeout 600 python3 tools/batch_sweep.py 400 32 64; done
for ks in 2 4; do DRNMF_SPLIT=4 DRNMF_KS=$ks timeout 600 python3 tools/batch_sweep.py 400 64; done
export DRNMF_SWEEP_SHAPE="257 1000 5"
timeout 600 python3 tools/batch_sweep.py 500 16 32 64
for ks in 1 2 4; do DRNMF_SPLIT=2 DRNMF_KS=$ks timeout 600 python3 tools/batch_sweep.py 500 32 64; done
} > "$OUT/sweep.txt" 2> "$OUT/sweep.err"
cat "$OUT/sweep.txt"
