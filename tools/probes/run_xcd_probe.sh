#!/bin/bash
cd "$(dirname "$0")"
out=../../gpurun_out/xcd_local_probe3.txt
: > $out
for cfg in "13 208" "14 224" "32 512" "4 64"; do
  for m in 0 3 8 2 9; do
    timeout 60 ./xcd_local_probe $cfg 4000 $m >> $out 2>&1
  done
done
grep -v participants $out
