#!/bin/bash
# sweep of tools/probes/chain_probe on the GPU box; output -> gpurun_out/chain_probe.txt
cd "$(dirname "$0")"
P=./chain_probe
out=../../gpurun_out/chain_probe.txt
mkdir -p ../../gpurun_out
: > $out
run() { timeout 60 $P "$@" >> $out 2>&1 || echo "  (exit $? for $*)" >> $out; }
for ch in 4 2 1; do
  for pl in 0 1; do
    for bar in 0 1; do
      run $ch $pl 0 $bar 0          # barrier alone
      run $ch $pl 1 $bar 0          # + exchange
      run $ch $pl 3 $bar 0          # + MFMAs
      run $ch $pl 7 $bar 0          # + dictionary stream, early prefetch
      run $ch $pl 7 $bar 1          # late prefetch on the polling wave
    done
  done
done
# timelines (chain 0, workgroup 0)
run 4 0 15 0 0
run 4 1 15 0 0
run 4 1 15 1 1
run 4 1 11 0 0
tail -n 200 $out
