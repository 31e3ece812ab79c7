#!/bin/bash
# Same-box A/B of measurement builds of libdrnmf.so on the headline training step:
#   tools/ab_builds.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the shipped build)
# Each variant is rebuilt (DRNMF_EXTRA_FLAGS) and timed with bench.py --no-extras --no-cpu-baseline;
# the list is walked twice (ABAB) so that drift shows.
out=gpurun_out/ab_builds.txt
: > $out
for rep in 1 2; do
  for v in "$@"; do
    export DRNMF_EXTRA_FLAGS="$v"
    python dr-nmf_amd/build.py > gpurun_out/ab_build.log 2>&1 || { echo "build failed: $v" >> $out; tail -5 gpurun_out/ab_build.log >> $out; continue; }
    python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/ab_line.json 2> gpurun_out/ab_err.log
    python - "$v" >> $out <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/ab_line.json').read().strip().splitlines()[-1])
    b = d['step_breakdown_ms']
    print("%-44s step %.1f ms  fwd %.1f  bptt_seq %.1f  batched %.1f  | inference fwd %.1f ms  loss %.6f" % (
        repr(sys.argv[1]), d['ms_per_step'], b['cell_forward_chain'], b['bptt_sequential_pass'],
        b['bptt_time_batched_weight_gradients'], d['forward']['ms_per_step'], d['loss_last']))
except Exception as e:
    print(repr(sys.argv[1]), "FAILED", e)
PY
  done
done
unset DRNMF_EXTRA_FLAGS
cat $out
