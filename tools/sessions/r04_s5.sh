#!/bin/bash
# round 4, GPU session 5: full GPU suite on the new build + sub-batch split sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s5
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
{
for s in 1 2 3 4; do
  DRNMF_SPLIT=$s timeout 900 python3 tools/batch_sweep.py 200 128 192 250 256 384 512 1024
done
timeout 900 python3 tools/batch_sweep.py 200 64 128 192 250 256 384 512 1024
} > "$OUT/split_sweep.txt" 2> "$OUT/split_sweep.err"
cat "$OUT/split_sweep.txt"; tail -3 "$OUT/split_sweep.err"
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1
tail -15 "$OUT/pytest_gpu.txt"
