#!/bin/bash
# A/B of the product (lean) library against the -DDRNMF_MEASURE build on ONE box: headline step, forward,
# B = 250 inference slab.  Run from the repo root on the GPU box:  bash tools/ab_aids.sh > gpurun_out/ab_no_aids.txt
set -u
line() {  # line <label> <bench args...>
  local label=$1; shift
  python3 bench.py "$@" --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
f = d.get('forward', {})
print('%-34s value %.0f frames/s, %.2f ms/step; forward %.0f frames/s, %.2f us per chain launch' % ('$label', d['value'], d['ms_per_step'], f.get('value', 0), f.get('roofline', {}).get('launch_us', 0)))"
}
for build in lean measure lean2; do
  if [ $build = measure ]; then DRNMF_MEASURE=1 python3 dr-nmf_amd/build.py --force > /dev/null 2>&1; export DRNMF_MEASURE=1; fi
  if [ $build = lean2 ]; then unset DRNMF_MEASURE; python3 dr-nmf_amd/build.py --force > /dev/null 2>&1; fi
  echo "== $build build (`python3 -c "import sys; sys.path.insert(0,'dr-nmf_amd'); import build; print(build._src_hash()[:16], ' '.join(f for f in build.FLAGS if f.startswith('-D')))"`)"
  line "headline B=64 T=2000 (5 steps)" --steps 5 --warmup 2
  line "forward B=250 T=400" --forward-only --batch 250 --frames 400 --steps 3 --warmup 1
done
