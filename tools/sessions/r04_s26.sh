#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s26
mkdir -p "$OUT"
cd "$ROOT"
python3 dr-nmf_amd/build.py > "$OUT/build.log" 2>&1
timeout 900 python3 -m pytest tests -m gpu -q -x -k "snmf or ista or mu or head" 2>&1 | tail -5 > "$OUT/tests.txt"
cat "$OUT/tests.txt"
{
for v in 0 1; do
  DRNMF_THIN=$v timeout 300 python3 tools/thin_ab.py 2>&1 | tail -1
  DRNMF_THIN=$v timeout 300 python3 tools/snmf_profile.py 32768 513 1000 20 2>&1 | grep "per iteration"
done
} > "$OUT/ab.txt" 2>&1
cat "$OUT/ab.txt"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/snmf" -o stats -- python3 "$ROOT/tools/snmf_profile.py" 32768 513 1000 20 > "$OUT/snmf_under_rocprof.txt" 2> "$OUT/snmf.err"
cp "$(find "$OUT/snmf" -name '*kernel_stats.csv' | head -1)" "$OUT/snmf_kernel_stats.csv" 2>/dev/null
find "$OUT/snmf" -name '*kernel_trace.csv' -delete; find "$OUT/snmf" -name "*.db" -delete
head -22 "$OUT/snmf_kernel_stats.csv" | cut -c1-160
