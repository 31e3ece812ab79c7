#!/bin/bash
# Same-box A/B of measurement builds on inference shapes (tools/profile_shape.py):
#   tools/ab_shapes.sh "<flags 1>" "<flags 2>" ... -- "B T F r K" "B T F r K" ...
out=gpurun_out/ab_shapes.txt
: > $out
variants=(); shapes=(); seen=0
for a in "$@"; do
  if [ "$a" = "--" ]; then seen=1; continue; fi
  if [ $seen = 0 ]; then variants+=("$a"); else shapes+=("$a"); fi
done
for rep in 1 2; do
  for v in "${variants[@]}"; do
    export DRNMF_EXTRA_FLAGS="$v"
    python dr-nmf_amd/build.py > gpurun_out/ab_build.log 2>&1 || { echo "build failed: $v" >> $out; continue; }
    for s in "${shapes[@]}"; do
      echo "[$v] $s: $(python tools/profile_shape.py $s 2>/dev/null | tail -1)" >> $out
    done
  done
done
unset DRNMF_EXTRA_FLAGS
cat $out
