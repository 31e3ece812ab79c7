#!/bin/bash
# in-kernel timeline of the single-XCD chain skeleton (features bit 4)
cd "$(dirname "$0")"
out=../../gpurun_out/xcd_chain_probe3.txt
mkdir -p ../../gpurun_out
: > $out
P=./xcd_chain_probe
for ch in 1 8; do
  for f in 16 24 17 19 23; do timeout 60 $P $ch 1 512 1 1 25 $f >> $out 2>&1; done
done
timeout 60 $P 8 2 512 1 1 25 16 >> $out 2>&1
timeout 60 $P 8 2 512 1 1 25 23 >> $out 2>&1
timeout 60 $P 8 2 512 2 2 25 23 >> $out 2>&1
cat $out
