#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trc3
rm -rf $OUT; mkdir -p $OUT
python3 $ROOT/scratch/train_c3.py 10 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $ROOT/scratch/train_c3.py 10 > /dev/null 2>&1
rm -f $OUT/s_kernel_trace.csv
head -32 $OUT/s_kernel_stats.csv | cut -c1-150
