#!/bin/bash
# barrier scaling with the number of concurrent single-XCD chains; agent- vs workgroup-scope sync atomics
cd "$(dirname "$0")"
out=../../gpurun_out/xcd_chain_probe2.txt
mkdir -p ../../gpurun_out
: > $out
for P in ./xcd_chain_probe ./xcd_chain_wg_probe; do
  echo "## $P" >> $out
  for ch in 1 2 4 8; do
    for f in 0 8 1 9; do timeout 60 $P $ch 1 512 1 1 25 $f >> $out 2>&1; done
  done
  timeout 60 $P 8 2 512 1 1 25 7 >> $out 2>&1
  timeout 60 $P 8 2 512 2 2 25 7 >> $out 2>&1
done
cat $out
