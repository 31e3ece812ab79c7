#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/c5pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--bins 1025 --r 4000 --layers 10 --frames 6 --batch 64 --no-cpu-baseline --no-ista --no-train --no-slab --steps 1 --warmup 0"
timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/f -o p -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/f.err
python3 $ROOT/profiles/summarize_pmc.py $OUT/s.json $OUT/f > /dev/null
python3 - <<PY
import json
d=json.load(open('$OUT/s.json'))
for k,v in d.items():
    if k.startswith('cell_'):
        print(k, {c:(round(x['mean']) if isinstance(x,dict) else round(x)) for c,x in v.items()})
PY
rm -rf $OUT/f
