#!/bin/bash
# Round evidence recipe; run from the repo root on the GPU box:
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r02a'
# Writes under gpurun_out/<tag>/ (scratch); copy what should be judged into profiles/.
# PMC passes are separate runs with --kernel-trace only (never with sys/hip/hsa traces); the
# program after `--` is python3 itself (no env/bash hop).
set -u
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
stats() {   # stats <name> <program args...>: rocprofv3 --kernel-trace --stats summary -> $OUT/<name>_kernel_stats.csv
    local name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o stats -- python3 "$@" \
        > "$OUT/${name}_under_rocprof.json" 2> "$OUT/$name.err"
    cp "$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${name}_kernel_stats.csv" 2>/dev/null
    find "$OUT/$name" -name '*kernel_trace.csv' -size +8M -delete
    find "$OUT/$name" -name "*.db" -delete
}
# 1. the default bench line (fwd+bwd headline, forward block, extras, CPU baseline)
python3 "$B" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
# 2. kernel trace + stats of the headline workload alone: one warm-up + one timed training step
#    (forward chain, head, loss, BPTT chain, time-batched weight gradients, Adam) and the forward
#    block; the extra lines reuse the same kernel templates at other shapes and would blur the means
stats bench "$B" --steps 1 --warmup 1 --no-cpu-baseline --no-extras
# 3. the forward alone (what the PMC passes below also run)
stats fwd "$B" --forward-only --steps 1 --warmup 1 --no-cpu-baseline --no-extras
# 4. frame-parallel ISTA GEMMs, the config-5 shape (fp32 and fp16 operands), the shipped training config
stats ista "$ROOT/tools/ista_profile.py" 32768 513 2000 25
stats c5 "$ROOT/tools/c5_profile.py" 16
stats train_c3 "$ROOT/tools/train_profile.py" 32 500 257 1000 5 3
# round 6: the same two workloads with the frame-parallel products in the split-operand mode (DRNMF_MATRIX_BF16X3)
stats ista_x3 "$ROOT/tools/ista_profile.py" 32768 513 2000 25 bf16x3
stats snmf_train_x3 "$ROOT/tools/snmf_profile.py" 32768 513 1000 20 both bf16x3
# ... the config-5 shape with fp16 operands ALONE (bench.py's config5_shape.f16.roofline.frac_rocprof reads this file)
stats c5_f16 "$ROOT/tools/profile_shape.py" 64 16 1025 4000 50 f16
# ... error of the mode against fp64 beside the exact-fp32 mode's, and the speed check
python3 "$ROOT/tools/x3_error_table.py" > "$OUT/x3_error_table.md" 2> "$OUT/x3_error_table.err"
python3 "$ROOT/tools/x3_check.py" > "$OUT/x3_check.txt" 2>&1
# ... SQ counters of the ISTA products in both modes (clock, matrix-pipe busy, LDS conflicts)
( cd "$ROOT" && bash tools/x3_pmc.sh "$TAG/x3pmc" "f32 bf16x3" > "$OUT/x3_pmc.txt" 2>&1 )
rm -rf "$OUT/x3pmc"
# dictionary training (sparse_nmf_gpu.m:210-298), 20 iterations each of KL and ED on 32768 x 513 x 1000
stats snmf_train "$ROOT/tools/snmf_profile.py" 32768 513 1000 20
python3 "$ROOT/tools/snmf_profile.py" 32768 513 1000 20 > "$OUT/snmf_train_unprofiled.txt" 2>&1
# the reference's inference slab (250 utterances) and B = 1024: forward only, sub-batches on side streams
stats fwd_b250 "$B" --forward-only --batch 250 --frames 200 --steps 1 --warmup 1 --no-cpu-baseline --no-extras
stats fwd_b1024 "$B" --forward-only --batch 1024 --frames 100 --steps 1 --warmup 1 --no-cpu-baseline --no-extras
python3 "$ROOT/tools/batch_sweep.py" 400 64 128 250 512 1024 2048 > "$OUT/batch_sweep.txt" 2> "$OUT/batch_sweep.err"
python3 "$ROOT/tools/batch_sweep.py" 2000 250 >> "$OUT/batch_sweep.txt" 2>> "$OUT/batch_sweep.err"
# HIP API calls of 20 steady-state optimiser steps (with-steps minus zero-steps trace): no synchronising call
for nst in 0 20; do
  timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d "$OUT/step_api_$nst" -o api -- python3 "$ROOT/tools/step_api_trace.py" 32 500 257 100 5 $nst \
      > "$OUT/step_api_${nst}.txt" 2> "$OUT/step_api_$nst.err"
  cp "$(find "$OUT/step_api_$nst" -name '*hip_api_stats.csv' | head -1)" "$OUT/step_hip_api_stats_$nst.csv" 2>/dev/null
done
python3 "$ROOT/tools/hip_api_delta.py" "$OUT/step_hip_api_stats_0.csv" "$OUT/step_hip_api_stats_20.csv" 20 > "$OUT/step_hip_api_delta.txt" 2>&1
find "$OUT" -name '*hip_api_trace.csv' -delete
# the other shipped dictionary size (r = 100): persistent Gram chains (cell_gram_persist.h), forward + BPTT
stats train_r100 "$ROOT/tools/train_profile.py" 32 500 257 100 5 10
# small-shape lines alone, unprofiled (C1 + the r = 100 / r = 1000 training steps), persistent chains on / off
python3 "$ROOT/tools/small_shapes.py" 20 > "$OUT/small_shapes.json" 2> "$OUT/small_shapes.err"
DRNMF_PERSIST=0 python3 "$ROOT/tools/small_shapes.py" 20 > "$OUT/small_shapes_persist0.json" 2>> "$OUT/small_shapes.err"
# N > 1 path of bench.py on the one GPU of this box: two ranks over gloo, torch all-reduce selected
# EXPLICITLY (RCCL refuses two ranks on one device); the only multi-rank run a 1-GPU box allows
DRNMF_BENCH_BACKEND=gloo DRNMF_BENCH_DEVICE=0 DRNMF_DP_BACKEND=torch timeout 900 python3 -m torch.distributed.run \
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 "$B" --gpus 2 --frames 400 \
    --steps 2 --warmup 1 --no-extras --no-cpu-baseline > "$OUT/n2_gloo_one_gpu.json" 2> "$OUT/n2_gloo_one_gpu.err"
# ... and EIGHT ranks on the one device through bench.py's own spawn_ranks (build lock, rendezvous, rank-0 JSON)
# (round 5: with the multi-rank extras -- configs[3] data-parallel training, configs[4] replicas -- at toy shapes, and rank 0's CPU baseline)
( cd "$ROOT" && DRNMF_BENCH_TINY=1 DRNMF_BENCH_BACKEND=gloo DRNMF_BENCH_DEVICE=0 DRNMF_DP_BACKEND=torch timeout 900 python3 bench.py --gpus 8 --batch 4 \
    --frames 8 --steps 2 --warmup 1 2> "$OUT/n8_gloo_one_gpu.err" | tail -1 > "$OUT/n8_gloo_one_gpu.json" )
# 5. PMC passes, 20 frames of the headline forward (cell + head only)
SMALL="--forward-only --frames 20 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_fetch" -o p -- python3 "$B" $SMALL \
    > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_write" -o p -- python3 "$B" $SMALL \
    > /dev/null 2> "$OUT/pmc_write.err"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -o p -- python3 "$B" $SMALL \
    > /dev/null 2> "$OUT/pmc_sq.err"
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_summary.json" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
# 5b. the same passes at the reference's inference slab (B = 250; two sub-batches on side streams)
SLAB="--forward-only --batch 250 --frames 20 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_fetch_b250" -o p -- python3 "$B" $SLAB \
    > /dev/null 2> "$OUT/pmc_fetch_b250.err"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_write_b250" -o p -- python3 "$B" $SLAB \
    > /dev/null 2> "$OUT/pmc_write_b250.err"
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq_b250" -o p -- python3 "$B" $SLAB \
    > /dev/null 2> "$OUT/pmc_sq_b250.err"
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_b250_summary.json" "$OUT/pmc_fetch_b250" "$OUT/pmc_write_b250" "$OUT/pmc_sq_b250"
# 5c. config 5 (F=1025, N=8000, K=50 untied, B=64, fp16 operands), 8 frames: L2-side traffic of the two cell kernels with
#     and without cell_a's prefetching wave (FETCH_SIZE counts Infinity-Cache hits too: it shows how often a layer's
#     dictionary CROSSES the fabric, not where from)
for pf in 1; do      # (DRNMF_PF=0 is a -DDRNMF_MEASURE aid since round 6: the shipped configuration only)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_HIT_sum -d "$OUT/pmc_c5_fetch_pf$pf" -o p -- python3 "$ROOT/tools/profile_shape.py" 64 8 1025 4000 50 f16 \
      > /dev/null 2> "$OUT/pmc_c5_fetch_pf$pf.err"
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d "$OUT/pmc_c5_write_pf$pf" -o p -- python3 "$ROOT/tools/profile_shape.py" 64 8 1025 4000 50 f16 \
      > /dev/null 2> "$OUT/pmc_c5_write_pf$pf.err"
  python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_c5_pf${pf}_summary.json" "$OUT/pmc_c5_fetch_pf$pf" "$OUT/pmc_c5_write_pf$pf"
done
# 6. the persistent chains of the shipped r = 100 training step: wave-cycle split and matrix-pipe time
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_r100" -o p -- python3 "$ROOT/tools/train_profile.py" 32 500 257 100 5 3 \
    > /dev/null 2> "$OUT/pmc_r100.err"
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/pmc_r100_summary.json" "$OUT/pmc_r100"
# keep the merge-back small (gpurun copies back at most 64 MiB): the raw per-dispatch directories go once they are summarised
find "$OUT" -maxdepth 1 -type d -name 'pmc_*' -exec rm -rf {} +
find "$OUT" -maxdepth 1 -type d -name 'step_api_*' -exec rm -rf {} +
find "$OUT" -name '*counter_collection.csv' -size +8M -delete
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
find "$OUT" -name "*.db" -delete
ls -laR "$OUT" | head -80
