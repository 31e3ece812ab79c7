#!/bin/bash
# sweep of tools/probes/xcd_chain_probe on the GPU box; output -> gpurun_out/xcd_chain_probe.txt
cd "$(dirname "$0")"
P=./xcd_chain_probe
out=../../gpurun_out/xcd_chain_probe.txt
mkdir -p ../../gpurun_out
: > $out
run() { timeout 60 $P "$@" >> $out 2>&1 || echo "  (exit $? for $*)" >> $out; }
echo "## B = 256 (8 chains x 32 rows), F = 512, K = 25 untied" >> $out
for kc in "1 1" "2 2"; do
  for f in 0 1 3 7; do run 8 2 512 $kc 25 $f; done
  run 8 2 512 $kc 1 7          # tied: the dictionary stays in the L2
done
echo "## B = 128 (8 chains x 16 rows)" >> $out
for kc in "1 1" "2 2"; do run 8 1 512 $kc 25 7; run 8 1 512 $kc 25 3; done
echo "## B = 64 (4 chains x 16 rows, 4 XCDs idle)" >> $out
for kc in "1 1" "2 2"; do run 4 1 512 $kc 25 7; run 4 1 512 $kc 25 3; run 4 1 512 $kc 25 0; done
echo "## B = 32 as 2 chains x 16 rows / B = 64 as 2 x 32 (F = 512)" >> $out
run 2 1 512 1 1 25 7; run 2 2 512 1 1 25 7; run 2 2 512 2 2 25 7
echo "## shipped training configuration r = 1000: F = 256, K = 5, B = 32 = 2 chains x 16 rows" >> $out
for kc in "2 1" "4 2"; do for f in 0 1 3 7; do run 2 1 256 $kc 5 $f; done; done
echo "## the same as ONE chain of 32 rows" >> $out
for kc in "2 1" "4 2"; do run 1 2 256 $kc 5 7; done
echo "## single chain, B = 16 (configs[0]-like row count at N = 2000)" >> $out
run 1 1 512 1 1 25 7; run 1 1 512 2 2 25 7
cat $out
