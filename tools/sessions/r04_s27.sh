#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s27
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 1200 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -8 > "$OUT/tests.txt"
cat "$OUT/tests.txt"
{
for v in 0 1; do
  DRNMF_THIN=$v timeout 300 python3 tools/thin_ab.py 2>&1 | tail -1
  DRNMF_THIN=$v timeout 300 python3 tools/snmf_profile.py 32768 513 1000 20 2>&1 | grep "per iteration"
done
timeout 300 python3 tools/train_profile.py 32 500 257 100 5 20 2>&1 | tail -1
timeout 300 python3 tools/train_profile.py 32 500 257 1000 5 10 2>&1 | tail -1
} > "$OUT/ab.txt" 2>&1
cut -c1-900 "$OUT/ab.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/r100" -o stats -- python3 "$ROOT/tools/train_profile.py" 32 500 257 100 5 12 > "$OUT/r100_under_rocprof.txt" 2> "$OUT/r100.err"
cp "$(find "$OUT/r100" -name '*kernel_stats.csv' | head -1)" "$OUT/r100_kernel_stats.csv" 2>/dev/null
find "$OUT/r100" -name '*kernel_trace.csv' -delete; find "$OUT/r100" -name "*.db" -delete
head -24 "$OUT/r100_kernel_stats.csv" | cut -c1-150
