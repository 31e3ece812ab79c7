#!/bin/bash
# second sweep: the per-producer-flag protocol (chain_kernel2) against the counter barrier
cd "$(dirname "$0")"
P=./chain_probe
out=../../gpurun_out/chain_probe2.txt
mkdir -p ../../gpurun_out
: > $out
run() { timeout 60 $P "$@" >> $out 2>&1 || echo "  (exit $? for $*)" >> $out; }
for ch in 4 2 1; do
  for pl in 0 1; do
    run $ch $pl 7 0 0 4900 0 0       # counter barrier, reference
    for part in 0 1; do
      run $ch $pl 0 0 0 4900 1 $part
      run $ch $pl 1 0 0 4900 1 $part
      run $ch $pl 3 0 0 4900 1 $part
      run $ch $pl 7 0 0 4900 1 $part
    done
  done
done
run 4 1 15 0 0 4900 1 1
run 4 0 15 0 0 4900 1 1
run 4 1 11 0 0 4900 1 1
run 4 1 9 0 0 4900 1 1
run 4 1 8 0 0 4900 1 1
grep -v census $out
