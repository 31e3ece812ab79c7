#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_s35
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_ista" -o p -- python3 "$ROOT/tools/ista_profile.py" 32768 513 2000 5 \
    > /dev/null 2> "$OUT/pmc_ista.err"
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_ista_summary.json" "$OUT/pmc_ista"
python3 - <<'PY'
import json,os
d=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT','.'),'gpurun_out/r04_s35/pmc_ista_summary.json')))
for k,v in d.items():
    if 'gemm' in k: print(k[:90], json.dumps(v)[:900])
PY
find "$OUT" -name '*counter_collection.csv' -size +4M -delete; find "$OUT" -name "*.db" -delete
