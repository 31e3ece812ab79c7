#!/bin/bash
# round 4, GPU session 2: in-kernel timeline of the cell kernels at large batches (timeline build)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s2
mkdir -p "$OUT"
cd "$ROOT"
export DRNMF_TIMELINE=1
python3 dr-nmf_amd/build.py --force > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
{
for b in 64 256 512; do
  echo "== B=$b"; timeout 300 python3 tools/timeline.py $b 24 513 1000 25
done
echo "== B=256 RB=1"; DRNMF_RB=1 timeout 300 python3 tools/timeline.py 256 24 513 1000 25
} > "$OUT/timeline.txt" 2> "$OUT/timeline.err"
cat "$OUT/timeline.txt"; tail -n 5 "$OUT/timeline.err"
