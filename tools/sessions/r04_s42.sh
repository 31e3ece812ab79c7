#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
python3 dr-nmf_amd/build.py > /dev/null 2>&1
DRNMF_RBA=4 timeout 600 python3 -m pytest tests/test_gpu_parity.py -q -x -k "large_batch_sub_batches or cell_forward_matches_oracle or model_predict" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
timeout 600 python3 tools/batch_sweep.py 400 128 250 512 1024
DRNMF_RBA=4 timeout 600 python3 tools/batch_sweep.py 400 128 250 512 1024
DRNMF_RBA=4 DRNMF_SPLIT=1 timeout 600 python3 tools/batch_sweep.py 400 250 512
DRNMF_RBA=4 DRNMF_SPLIT=2 timeout 600 python3 tools/batch_sweep.py 400 512 1024
DRNMF_RBA=4 DRNMF_KS=1 timeout 600 python3 tools/batch_sweep.py 400 250 1024
