#!/usr/bin/env python3
"""Benchmark of the DR-NMF hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N ...; a bare `python bench.py --gpus N` starts exactly that
          as a child process before anything touches the GPU and relays its output)

Metric (BASELINE.json): STFT frames/sec (fwd+bwd), 513-bin x 2000-frame, K=25 unrolls.
Workload (BASELINE.json configs[1] shape): dictionary 513x2000 (1000 speech + 1000 noise atoms,
log_D / log_alph untied per layer as the shipped configs do), synthetic 513-bin x 2000-frame
spectrograms, 64 utterances per GPU, inputs resident in HBM.  One step = forward (recurrent cell +
mask head) + loss + BPTT + Adam over the batch (`train_on_batch`); with N > 1 GPUs every rank holds
64 utterances of its own (weak scaling) and the step contains the ONE collective of the path, the
all-reduce of the flat gradient (RCCL through libdrnmf's C ABI).  The forward alone -- the
north-star target and round 1's headline -- is timed the same way and reported beside it as
`forward` with its own roofline block.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense fp32
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--bins", type=int, default=513)
    ap.add_argument("--r", type=int, default=1000, help="atoms per source (N = 2r)")
    ap.add_argument("--layers", type=int, default=25)
    ap.add_argument("--tied", action="store_true", help="tie log_D/log_alph across layers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ista", action="store_true", help="skip the frame-parallel ISTA line")
    ap.add_argument("--operand-f16", action="store_true",
                    help="forward block with fp16 MFMA operands, fp32 accumulate (BASELINE config 5 "
                         "mode; the headline metric is quoted in fp32)")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the config-5 shape lines (F=1025, N=8000, K=50)")
    ap.add_argument("--no-slab", action="store_true",
                    help="skip the 250-utterance inference slab line")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training step at the shipped configuration (configs[2])")
    ap.add_argument("--forward-only", action="store_true",
                    help="time the forward only (value = forward frames/s, metric says so): for "
                         "profiling passes and boxes without the ~75 GB the BPTT needs")
    ap.add_argument("--no-extras", action="store_true", help="headline + forward blocks only")
    ap.add_argument("--cpu-frames", type=int, default=0,
                    help="frame-steps of the CPU baseline sample (0 = auto, ~15-25 s)")
    return ap.parse_args()


def synth_on_device(torch, dev, B, T, F, r, seed, want_clean=False):
    """SURVEY.md section 8d generator; the dictionary comes from numpy PCG64(7654), the
    activations/noise from torch's generator on the device (256M draws).  want_clean also
    returns the clean target Y = Htrue[:, :r] W[:, :r]^T."""
    N = 2 * r
    rng = np.random.Generator(np.random.PCG64(7654))
    W = rng.random((F, N)) ** 4
    W = (W / np.sqrt(np.sum(W * W, axis=0, keepdims=True))).astype(np.float32)
    log_h0 = rng.uniform(-0.05, 0.05, N).astype(np.float32)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    Wt = torch.from_numpy(W).to(dev)
    X = torch.empty((B, T, F), dtype=torch.float32, device=dev)
    Y = torch.empty((B, T, F), dtype=torch.float32, device=dev) if want_clean else None
    for b in range(B):
        Ht = (torch.rand((T, N), generator=g, device=dev) < 0.02) * \
            torch.rand((T, N), generator=g, device=dev) * 5.0
        X[b] = Ht @ Wt.t() + 0.01 * torch.rand((T, F), generator=g, device=dev)
        if want_clean:
            Y[b] = Ht[:, :r] @ Wt[:, :r].t()
    if want_clean:
        return W, log_h0, X, Y
    return W, log_h0, X


def ista_bench(torch, dev, F, N, K, W, n=32768):
    """Frame-parallel K-iteration ISTA-ED (enhance.py:402-418) on n frames: the MFMA-bound sibling
    of the recurrent cell (no dependence between frames).  4*F*N flops per frame per iteration."""
    from drnmf_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    Wt = torch.from_numpy(W).to(dev)
    Ht = (torch.rand((n, N), generator=g, device=dev) < 0.02) * \
        torch.rand((n, N), generator=g, device=dev) * 5.0
    Xf = Ht @ Wt.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
    H = torch.full((n, N), 0.1, dtype=torch.float32, device=dev)
    ops.ista_forward(Xf, Wt, H, 1.0, 400.0, 2)        # warm-up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.ista_forward(Xf, Wt, H, 1.0, 400.0, K)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    tf = n * 4.0 * F * N * K / sec / 1e12
    return {"frames": n, "K": K, "frames_per_s": n / sec, "tflops": tf,
            "frac_of_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS,
            "gemm_launch_us": sec / (2 * K) * 1e6}


def both_modes(fn, pick, *args, **kw):
    """fn(*args) with the frame-parallel products in the exact-fp32 mode (the figures at the top level, as in
    earlier rounds) and again in the split-operand mode (DRNMF_MATRIX_BF16X3, include/drnmf.h) under "bf16x3";
    `pick(result)` -> the time-like figure whose ratio is reported as speedup_bf16x3."""
    from drnmf_amd import ops
    prev = ops.set_matrix_mode("f32")
    try:
        res = fn(*args, **kw)
        ops.set_matrix_mode("bf16x3")
        alt = fn(*args, **kw)
    finally:
        ops.set_matrix_mode(prev)
    res["matrix_mode"] = "f32 (v_mfma_f32_32x32x2_f32, exact)"
    alt["matrix_mode"] = "f32 via bf16x3 operands, f32 accumulate (6 x v_mfma_f32_32x32x16_bf16 per product)"
    res["bf16x3"] = alt
    try:
        res["speedup_bf16x3"] = pick(res) / pick(alt)
    except Exception:        # noqa: BLE001
        pass
    return res


def ragged_inference_bench(torch, dev, F=257, r=1000, K=5, T=2000, n=2000, slab=250):
    """The reference's inference over a RAGGED data set (enhance.py:1181-1203: every utterance padded to the
    longest, masks cropped afterwards): `predict` as the reference runs it -- every slab at T_max, input order
    -- against the length-aware `predict` (utterances sorted by valid length, each slab at its own length,
    dr-nmf_amd/layers.py).  Lengths as SURVEY.md 8(d) draws them for the training configs: uniform in
    [0.4 T, T]; 2000 utterances (CHiME2's test / validation lists hold 1980 / 2460) = eight slabs of 250.  The
    figure is VALID frames per second, host arrays in and out (PCIe-inclusive: never `value`), the shipped
    r = 1000, K = 5 model at the reference's frame cap (util.py:319).  The two runs group different utterances into
    a slab, and a 250-row slab runs as sub-batches with their own kernel variants: equal to fp32 rounding, not bit
    for bit (bit-equality at one kernel variant: tests/test_gpu_parity.py)."""
    import time
    from drnmf_amd import layers
    N = 2 * r
    W, log_h0, X = synth_on_device(torch, dev, slab, T, F, r, seed=123)
    rng = np.random.Generator(np.random.PCG64(7654))
    lens = rng.integers(int(0.4 * T), T + 1, size=n)
    xh = np.concatenate([X.cpu().numpy()] * ((n + slab - 1) // slab))[:n].copy()
    for i, L in enumerate(lens):
        xh[i, L:] = -1.0
    del X
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
             alph=400.0 if r >= 1000 else 50.0, lam1=1.0, params_trainable=["log_D", "log_alph"],
             params_untied=["log_D", "log_alph"])
    model = layers.build_unfolded_snmf(p, device=dev)
    model.cell.log_h0.copy_(torch.from_numpy(log_h0))
    valid = int(lens.sum())
    res = {}
    ref = None
    for name, kw in (("padded_to_T_max", dict(length_aware=False)), ("length_aware", dict()),
                     ("length_aware_lengths_given", dict(lengths=lens))):
        model.predict(xh, batch_size=slab, **kw)           # staging buffers, graphs
        t0 = time.perf_counter()
        m = model.predict(xh, batch_size=slab, **kw)
        sec = time.perf_counter() - t0
        res[name] = {"valid_frames_per_s": valid / sec, "s": sec}
        if ref is None:
            ref = m
        else:
            res[name]["max_abs_diff_to_padded_run_on_valid_frames"] = float(
                max(np.abs(m[i, :L] - ref[i, :L]).max() for i, L in enumerate(lens) if L))
        del m
    res.update({"utterances": n, "T_max": T, "F": F, "N": N, "K": K, "slab": slab, "valid_frames": valid,
                "mean_length_over_T_max": float(lens.mean() / T),
                "speedup_valid_frames": res["length_aware"]["valid_frames_per_s"] /
                res["padded_to_T_max"]["valid_frames_per_s"]})
    model.free_predict_buffers()
    del model, xh, ref
    torch.cuda.empty_cache()
    return res


def mu_bench(torch, dev, F, N, W, n=32768, iters=20):
    """SNMF inference by multiplicative updates with W fixed -- the classical baseline branch of
    enhance.py:838-852 that replaces the Matlab process (200 iterations there) -- on n frames,
    beta = 2.  Flops counted as the Matlab loop spends them: 6*F*N per frame per iteration."""
    from drnmf_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    Wt = torch.from_numpy(W).to(dev)
    Ht = (torch.rand((n, N), generator=g, device=dev) < 0.02) * \
        torch.rand((n, N), generator=g, device=dev) * 5.0
    V = Ht @ Wt.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
    H = torch.rand((n, N), generator=g, device=dev)
    ops.mu_forward(V, Wt, H, 5.0, 2, beta=2.0)            # warm-up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.mu_forward(V, Wt, H, 5.0, iters, beta=2.0)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    tf = n * 6.0 * F * N * iters / sec / 1e12
    return {"frames": n, "iterations": iters, "frame_iterations_per_s": n * iters / sec,
            "tflops_as_counted_by_the_reference_loop": tf,
            "ms_per_iteration": sec / iters * 1e3}


def snmf_train_bench(torch, dev, F, r, n=32768, iters=20):
    """Dictionary training (sparse_nmf_gpu.m:156-298 through snmf.py's loop): iterations per second of
    the W + H multiplicative updates on n frames, r atoms, beta = 1 (the reference's 'kl' default) and
    beta = 2, objective logged on the device every iteration, no host synchronisation inside the
    loop (conv_eps = 0, as enhance.py:842 and the shipped configs set it)."""
    from drnmf_amd import ops
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    V = torch.rand((n, F), generator=g, device=dev) ** 2 + 1e-3
    out = {"frames": n, "F": F, "atoms": r, "iterations": iters}
    for name, beta in (("kl", 1.0), ("ed", 2.0)):
        W0 = torch.rand((F, r), generator=g, device=dev)
        H0 = torch.rand((n, r), generator=g, device=dev)
        tr = ops.SnmfTrainer(V, W0, H0, beta=beta)
        log = torch.zeros((iters + 2, 2), dtype=torch.float32, device=dev)
        for i in range(2):
            tr.step(5.0, None, True, obj=log[i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            tr.step(5.0, None, True, obj=log[2 + i])
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3
        c = log[:, 1].cpu().numpy()
        out[name] = {"iterations_per_s": iters / sec, "ms_per_iteration": sec / iters * 1e3,
                     "cost_first": float(c[0]), "cost_last": float(c[-1])}
        del tr, W0, H0
    del V
    torch.cuda.empty_cache()
    return out


def train_bench(torch, dev, steps=20, warmup=5, shape=(32, 500, 257, 1000, 5), ragged=True):
    """Forward + BPTT + Adam per step on a synthetic batch.  Default shape = BASELINE configs[2],
    the shipped training configuration (downsample1: F=257, maxlen=500, batch 32, K=5, r=1000,
    untied log_D/log_alph, ragged lengths); bench also runs it at the headline shape (F=513,
    T=2000, K=25, B=64) for the metric's "fwd+bwd" reading.  Algorithmic flops fwd+bwd =
    12*F*N*K - 2*F*N per frame."""
    from drnmf_amd import layers
    B, T, F, r, K = shape
    N = 2 * r
    W, _, x, y = synth_on_device(torch, dev, B, T, F, r, seed=7654, want_clean=True)
    w = torch.ones((B, T), dtype=torch.float32, device=dev)
    if ragged:      # lengths in [0.4 T, T], -1 padding after the valid prefix (audio_dataset.py:144)
        lens = np.random.Generator(np.random.PCG64(7654)).integers(int(0.4 * T), T + 1, size=B)
        keep = torch.arange(T, device=dev)[None, :] < torch.from_numpy(lens).to(dev)[:, None]
        w = keep.to(torch.float32)
        x = torch.where(keep[..., None], x, torch.full_like(x, -1.0))
        y = torch.where(keep[..., None], y, torch.full_like(y, -1.0))
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=W, alph=400.0 if r >= 1000 else 50.0, lam1=1.0,
             params_untied=["log_D", "log_alph"], params_trainable=["log_D", "log_alph"])
    model = layers.build_unfolded_snmf(p, device=dev)
    model.compile(lr=1e-3)
    losses = [model.train_on_batch(x, y, w) for _ in range(max(warmup, 1))]   # graph build + warm-up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        losses.append(model.train_on_batch(x, y, w))
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / steps
    model.backward_profile = {}
    model.phase_events = {}
    model.train_on_batch(x, y, w)
    bp, pe = model.backward_profile, model.phase_events
    fwd_ms = pe["cell_forward"][0].elapsed_time(pe["cell_forward"][1])
    from drnmf_amd import ops as _ops
    lpf = _ops.cell_launches_per_frame(model.cell._desc(B, T))     # 2K-1 factored, K-1 Gram form
    launches = T * lpf + bp["chain_launches"]
    launch_us = (fwd_ms + bp["chain_ms"]) * 1e3 / launches
    valid = float(w.sum().item())
    flops = (12.0 * F * N * K - 2.0 * F * N) * B * T
    del model, x, y, w
    torch.cuda.empty_cache()
    return {"config": "F=%d N=%d K=%d B=%d T=%d untied, %s (%.0f%% valid)" %
                      (F, N, K, B, T, "ragged" if ragged else "full length",
                       100.0 * valid / (B * T)),
            "ms_per_step": sec * 1e3, "frames_per_s": B * T / sec,
            "valid_frames_per_s": valid / sec, "tflops": flops / sec / 1e12,
            "frac_of_f32_mfma_peak": flops / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "chain_launches_per_step": launches, "chain_launch_us": launch_us,
            "form": "gram (one B x N x N contraction per layer-step)" if lpf == K - 1
                    else "factored (two B x F x N contractions per layer-step)",
            "cell_forward_ms": fwd_ms, "bptt_sequential_ms": bp["chain_ms"],
            "bptt_time_batched_ms": bp["batched_ms"], "steps": steps, "warmup": warmup,
            "timing": "HIP events",
            "loss_first": float(losses[0]), "loss_last": float(losses[-1])}


def slab_bench(torch, dev, F, r, K, T, slab=250, host_slabs=0):
    """The reference predicts in slabs of 250 utterances (enhance.py:1189-1193): the same forward
    (cell + mask head) on one slab, where the row-blocked kernels apply."""
    from drnmf_amd import layers, ops
    N = 2 * r
    W, log_h0, X = synth_on_device(torch, dev, slab, T, F, r, seed=99)
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
             alph=400.0 if r >= 1000 else 50.0, lam1=1.0, params_trainable=["log_D", "log_alph"],
             params_untied=["log_D", "log_alph"])
    model = layers.build_unfolded_snmf(p, device=dev)
    h_buf = torch.empty((slab, T, N), dtype=torch.float32, device=dev)
    m_buf = torch.empty((slab, T, F), dtype=torch.float32, device=dev)

    def step():
        h = model.cell.call(X, mask_value=-1., out=h_buf)
        ops.head_forward(h, model.clean.kernel, model.noise.kernel, out=m_buf)
    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    tf = slab * T * 4.0 * F * N * K / sec / 1e12
    out = {"utterances": slab, "frames_per_s": slab * T / sec, "ms_per_slab": sec * 1e3,
           "tflops": tf, "frac_of_f32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS}
    if host_slabs:
        # host arrays in, host arrays out (never `value`): the reference's loop of predict_on_batch calls
        # (enhance.py:1189-1193) against model.predict, whose copies ride two side streams
        import time
        import numpy as np
        xh = np.concatenate([X.cpu().numpy()] * host_slabs)
        model.predict(xh[:slab + 1], batch_size=slab)              # both sets of staging buffers, pinned pages
        t0 = time.perf_counter()
        kept = [model.predict_on_batch(xh[s0:s0 + slab]) for s0 in range(0, xh.shape[0], slab)]
        t_loop = time.perf_counter() - t0
        del kept                                                   # (unmapping GBs of results is not timed)
        t0 = time.perf_counter()
        kept = model.predict(xh, batch_size=slab)
        t_pipe = time.perf_counter() - t0
        del kept
        out["host_arrays_in_and_out"] = {
            "slabs": host_slabs, "frames_per_s_predict_on_batch_loop": xh.shape[0] * T / t_loop,
            "frames_per_s_predict": xh.shape[0] * T / t_pipe,
            "note": "PCIe-inclusive, wall clock, pageable numpy in and out, result allocation included; "
                    "both stage through pinned buffers, predict also sends slab s+1 while slab s computes"}
        del xh
    del model, X, h_buf, m_buf
    torch.cuda.empty_cache()
    return out


def c5_pmc_traffic():
    """L2-side bytes per launch of the config-5 fp16 chain kernels from the newest committed PMC summary
    (profiles/r*_pmc_c5_pf1_summary.json; FETCH_SIZE x 2 + WRITE_SIZE as in pmc_traffic), per kernel."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_c5_pf1_summary.json")))
    if not files:
        return None
    try:
        js = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    per = {k: v.get("hbm_side_read_bytes_per_launch", 0.0) + v.get("hbm_side_write_bytes_per_launch", 0.0)
           for k, v in js.items() if isinstance(v, dict) and k.startswith("cell_") and
           "hbm_side_read_bytes_per_launch" in v}
    return {"file": "profiles/" + os.path.basename(files[-1]), "bytes_per_launch_by_kernel": per,
            "lib_src_sha16_of_profile": js.get("_lib_src_sha16")}


def config5_bench(torch, dev, frames=64, B=64):
    """BASELINE configs[4] shape on ONE GPU (the config itself is an 8-GPU stress line): W 1025 x
    8000, K=50 untied, forward (cell + head) with fp32 and with fp16 MFMA operands (fp32
    accumulate).  Flop fractions against the fp32 / fp16 dense MFMA peaks respectively."""
    from drnmf_amd import layers, ops
    F, r, K = 1025, 4000, 50
    N = 2 * r
    W, log_h0, X = synth_on_device(torch, dev, B, frames, F, r, seed=5)
    out = {"shape": "F=%d N=%d K=%d B=%d T=%d untied" % (F, N, K, B, frames)}
    h_buf = torch.empty((B, frames, N), dtype=torch.float32, device=dev)
    m_buf = torch.empty((B, frames, F), dtype=torch.float32, device=dev)
    masks = {}
    for name, od, peak in (("f32", "float32", PEAK_F32_MFMA_TFLOPS), ("f16", "float16", 2500.0)):
        p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=frames,
                 K_layers=K, W=W, alph=0.4 * r, lam1=1.0, params_trainable=["log_D", "log_alph"],
                 params_untied=["log_D", "log_alph"], operand_dtype=od)
        model = layers.build_unfolded_snmf(p, device=dev)
        model.cell.log_h0.copy_(torch.from_numpy(log_h0))

        def step():
            h = model.cell.call(X, mask_value=-1., out=h_buf)
            ops.head_forward(h, model.clean.kernel, model.noise.kernel, out=m_buf)
        step()
        torch.cuda.synchronize()
        secs = []
        for _ in range(3):                    # best of three single passes (one pass is ~70-120 ms)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            torch.cuda.synchronize()
            secs.append(e0.elapsed_time(e1) * 1e-3)
        sec = min(secs)
        tf = B * frames * 4.0 * F * N * K / sec / 1e12
        launch_us = sec / (frames * (2 * K - 1)) * 1e6
        # HBM-bound at this shape (K = 50 untied dictionaries = 1.6 GB fp16 / 3.3 GB fp32, past the
        # 256 MB Infinity Cache): algorithmic bytes per launch = the layer's dictionary once (one
        # packing) + the activations it exchanges (h in, residual out, or the reverse)
        esz = 2 if name == "f16" else 4
        bytes_launch = float(F) * N * esz + B * (N + F) * esz
        gbs = bytes_launch / (launch_us * 1e-6) / 1e9
        # ... and priced per LAYER-STEP: a layer's dictionary is algorithmically needed once for the
        # cell_b / cell_a pair that uses it (fp16: one packing, read from HBM once -- by cell_a's prefetching
        # wave of the layer before -- and then out of the L2 / the Infinity Cache; fp32: two packings)
        step_us = sec / (frames * K) * 1e6
        bytes_step = float(F) * N * esz + 2.0 * B * (N + F) * esz
        gbs_step = bytes_step / (step_us * 1e-6) / 1e9
        # the same launch under the kernel trace (committed: tools/profile_shape.py 64 16 1025 4000 50 f16)
        prof = rocprof_launch_us("r*_c5_f16_kernel_stats.csv") if name == "f16" else None
        out[name] = {"frames_per_s": B * frames / sec, "tflops": tf, "frac_of_mfma_peak": tf / peak,
                     "launch_us": launch_us, "layer_step_us": step_us,
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                  "bytes_per_launch": bytes_launch,
                                  "algorithmic_bytes_per_launch": bytes_launch,
                                  "launch_us_rocprof": prof,
                                  "frac_rocprof": (bytes_launch / (prof["mean_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS)
                                  if prof else None,
                                  "traffic": c5_pmc_traffic() if name == "f16" else None},
                     "roofline_per_layer_step": {"bound": "hbm", "achieved": gbs_step, "peak": PEAK_HBM_GBS,
                                                 "unit": "GB/s", "frac": gbs_step / PEAK_HBM_GBS,
                                                 "bytes_per_layer_step": bytes_step}}
        masks[name] = m_buf.clone()
        del model
    out["mask_mse_f16_vs_f32"] = float(((masks["f16"] - masks["f32"]) ** 2).mean())
    del X, h_buf, m_buf, masks
    torch.cuda.empty_cache()
    return out


def dp_train_bench(torch, dev, dist, world, rank, shape, steps=10, warmup=3):
    """BASELINE configs[3]: data-parallel training, every rank B utterances of its own (global batch =
    world x B; 8 x 32 = 256 at the shipped shape), the PRODUCT's train_on_batch -- forward + loss + BPTT on
    the local shard, ONE all-reduce of the flat gradient buffer over the library's RCCL communicator
    (drnmf_allreduce_grads), fused Adam -- ragged lengths, weights broadcast from rank 0 by compile().
    A COLLECTIVE function: every rank calls it.  Timed like the headline (barrier + synchronize on both
    sides, max over ranks); frames/s is summed over ranks (weak scaling)."""
    from drnmf_amd import layers, dp
    B, T, F, r, K = shape
    N = 2 * r
    W, _, x, y = synth_on_device(torch, dev, B, T, F, r, seed=7654 + rank, want_clean=True)
    lens = np.random.Generator(np.random.PCG64(7654 + rank)).integers(int(0.4 * T), T + 1, size=B)
    keep = torch.arange(T, device=dev)[None, :] < torch.from_numpy(lens).to(dev)[:, None]
    w = keep.to(torch.float32)
    x = torch.where(keep[..., None], x, torch.full_like(x, -1.0))
    y = torch.where(keep[..., None], y, torch.full_like(y, -1.0))
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=W, alph=400.0 if r >= 1000 else 50.0, lam1=1.0,
             params_untied=["log_D", "log_alph"], params_trainable=["log_D", "log_alph"])
    model = layers.build_unfolded_snmf(p, device=dev)
    model.compile(lr=1e-3)                        # (broadcasts rank 0's weights)
    losses = [model.train_on_batch(x, y, w) for _ in range(max(warmup, 1))]
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(model.train_on_batch(x, y, w))
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    flat = model._flat
    acc = torch.tensor([wall, float(w.sum().item())], dtype=torch.float64, device=dev)
    tmax = acc[:1].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    vsum = acc[1:].clone()
    dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
    sec = float(tmax.item()) / steps
    # the step's one collective alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dp.allreduce_sum_(flat)
    torch.cuda.synchronize()
    dist.barrier()
    e0.record()
    for _ in range(5):
        dp.allreduce_sum_(flat)
    e1.record()
    torch.cuda.synchronize()
    ar_ms = e0.elapsed_time(e1) / 5.0
    res = {"config": "F=%d N=%d K=%d untied, %d utterances x %d frames per GPU, ragged; global batch %d" %
                     (F, N, K, B, T, world * B),
           "n_gpus": world, "ms_per_step": sec * 1e3, "frames_per_s": world * B * T / sec,
           "valid_frames_per_s": float(vsum.item()) / sec,
           "allreduce_bytes": int(flat.numel()) * 4, "allreduce_ms": ar_ms,
           "allreduce_GBps_bus": (2.0 * (world - 1) / world) * flat.numel() * 4 / (ar_ms * 1e-3) / 1e9,
           "steps": steps, "warmup": warmup, "timing": "wall clock, barrier + synchronize, max over ranks",
           "loss_first": float(losses[0]), "loss_last": float(losses[-1])}
    del model, x, y, w
    torch.cuda.empty_cache()
    return res


def c5_replicas_bench(torch, dev, dist, world, rank, frames=16, B=64, shape=(1025, 4000, 50)):
    """BASELINE configs[4] on N GPUs: the path has no exchange step in inference -- every rank runs the
    config-5 forward (fp16 MFMA operands, fp32 accumulate; cell + mask head) on its own B utterances
    ("replicas only").  COLLECTIVE (barriers).  frames/s summed over ranks; the
    per-launch figure is the slowest rank's."""
    from drnmf_amd import layers, ops
    F, r, K = shape
    N = 2 * r
    W, log_h0, X = synth_on_device(torch, dev, B, frames, F, r, seed=5 + rank)
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=frames, K_layers=K, W=W,
             alph=0.4 * r, lam1=1.0, params_trainable=["log_D", "log_alph"],
             params_untied=["log_D", "log_alph"], operand_dtype="float16")
    model = layers.build_unfolded_snmf(p, device=dev)
    model.cell.log_h0.copy_(torch.from_numpy(log_h0))
    h_buf = torch.empty((B, frames, N), dtype=torch.float32, device=dev)
    m_buf = torch.empty((B, frames, F), dtype=torch.float32, device=dev)

    def step():
        h = model.cell.call(X, mask_value=-1., out=h_buf)
        ops.head_forward(h, model.clean.kernel, model.noise.kernel, out=m_buf)
    step()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        dist.barrier()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        best = float(tt.item()) if best is None else min(best, float(tt.item()))
    ok = bool(torch.isfinite(m_buf).all().item())
    launch_us = best / (frames * (2 * K - 1)) * 1e6
    res = {"shape": "F=%d N=%d K=%d untied, fp16 operands, %d utterances x %d frames per GPU" % (F, N, K, B, frames),
           "n_gpus": world, "frames_per_s": world * B * frames / best, "launch_us_slowest_rank": launch_us,
           "GBps_per_gpu_algorithmic": (float(F) * N * 2 + B * (N + F) * 2) / (launch_us * 1e-6) / 1e9,
           "finite_masks": ok, "timing": "wall clock, barrier on both sides, max over ranks, best of 3"}
    del model, X, h_buf, m_buf
    torch.cuda.empty_cache()
    return res


def dense_graph_bench(torch, dev, F, r, K, B, frames=40):
    """The reference's op graph as written -- relu(p U_k + h S_k + x Wk_k + b_k), dense U and
    materialised Gram S (custom_layers.py:361-369; what cpu_baseline times on the host) -- on the
    general dense-matrix kernel (csrc/cell_dense.hip), same shape as the headline.  Flops are the
    ones this form executes, 2*B*(K*(N+F)+(K-1)*N)*N per frame; the algorithmic count of the
    headline (4*F*N*K) is what `frames_per_s` should be compared on."""
    from drnmf_amd import ops
    N = 2 * r
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    U = torch.randn((K, N, N), generator=g, device=dev) * (0.5 / N ** 0.5)
    S = torch.randn((K - 1, N, N), generator=g, device=dev) * (0.5 / N ** 0.5)
    W = torch.randn((K, F, N), generator=g, device=dev) * (0.5 / F ** 0.5)
    b = torch.zeros((K, N), device=dev)
    X = torch.rand((B, frames, F), generator=g, device=dev)
    h0 = torch.zeros(N, device=dev)
    desc = ops.make_dense_desc(B, frames, F, N, K)
    P = ops.dense_prepare_params(desc, U, S if K > 1 else None, W, b)
    ws = ops.dense_workspace(desc, dev)
    out = torch.empty((B, frames, N), device=dev)
    ops.dense_cell_forward(X, None, P, desc, h0, out=out, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.dense_cell_forward(X, None, P, desc, h0, out=out, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    fl = 2.0 * B * frames * N * (K * (N + F) + (K - 1) * N)
    bytes_ls = ((K * (N + F) + (K - 1) * N) * N / K) * 4.0      # matrix bytes streamed per layer-step (mean)
    res = {"frames": frames, "frames_per_s": B * frames / sec,
           "us_per_layer_step": sec / (frames * K) * 1e6, "tflops_executed": fl / sec / 1e12,
           "tflops_algorithmic": B * frames * 4.0 * F * N * K / sec / 1e12,
           "matrix_stream_GBps": bytes_ls / (sec / (frames * K)) / 1e9}
    # the same with fp16-stored matrices (drnmf_dense_desc_t.operand_f16: half the stream)
    d16 = ops.make_dense_desc(B, frames, F, N, K, operand_f16=True)
    P16 = ops.dense_prepare_params(d16, U, S if K > 1 else None, W, b)
    ops.dense_cell_forward(X, None, P16, d16, h0, out=out, workspace=ws)
    torch.cuda.synchronize()
    e0.record()
    ops.dense_cell_forward(X, None, P16, d16, h0, out=out, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    s16 = e0.elapsed_time(e1) * 1e-3
    res["f16_operands"] = {"frames_per_s": B * frames / s16, "us_per_layer_step": s16 / (frames * K) * 1e6,
                           "matrix_stream_GBps": 0.5 * bytes_ls / (s16 / (frames * K)) / 1e9}
    del U, S, W, P, P16, ws, out, X
    torch.cuda.empty_cache()
    return res


def stft_bench(torch, dev, n_sig=64, seconds=10.0, N=1024, hop=256, reps=10):
    """The STFT-magnitude front end alone: white-noise int16 PCM, 16 kHz, 10 s per
    signal -> |STFT| (util.py:171-201, audio_dataset.py:22-23,194).  HBM-bound: bytes = PCM in +
    magnitudes out."""
    from drnmf_amd import ops
    nsampl = int(16000 * seconds)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    pcm = torch.randint(-20000, 20000, (n_sig, nsampl), generator=g, device=dev,
                        dtype=torch.int32).to(torch.int16)
    mag = ops.stft_mag(pcm, N=N, hop=hop)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mag = ops.stft_mag(pcm, N=N, hop=hop)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    frames = mag.shape[0] * mag.shape[1]
    nbytes = pcm.numel() * 2 + mag.numel() * 4
    return {"signals": n_sig, "seconds_each": seconds, "N_fft": N, "hop": hop,
            "frames_per_s": frames / sec, "us": sec * 1e6, "GBps_in_plus_out": nbytes / sec / 1e9,
            "frac_of_8TBps": nbytes / sec / 8e12}


def run_guarded(fn, timeout_s, on_timeout):
    """fn() with a watchdog thread: exceptions become {'error': ...}; if fn has not returned after
    timeout_s the watchdog calls on_timeout() (which is expected to end the process)."""
    import threading
    done = threading.Event()

    def watchdog():
        if not done.wait(timeout_s):
            on_timeout()
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        res = fn()
    except Exception as e:           # noqa: BLE001 -- reported in the JSON line instead
        res = {"error": repr(e)[:300]}
    done.set()
    return res


def pmc_traffic():
    """HBM-side bytes per cell launch from the newest committed PMC summary (separate rocprofv3
    --pmc passes over this same command, profiles/collect.sh; FETCH_SIZE KB x 1024 x 2 -- the
    gfx950 under-report of wide coalesced reads, MI355X_MICROARCH.md -- plus WRITE_SIZE KB x
    1024).  PMC counters cannot be read from inside the process, so this is the PROFILED figure of
    the same kernels.  Returns (traffic, from_profile): `traffic` is the figure only when the
    summary was collected on a library built from exactly the sources this run uses (content hash
    `_lib_src_sha16`, stamped by profiles/summarize_pmc.py), else None; `from_profile` always says
    what the committed summary holds and for which sources."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_summary.json")))
    try:
        sys.path.insert(0, os.path.join(here, "dr-nmf_amd"))
        import build as _b
        cur = _b._src_hash()[:16]
    except Exception:            # noqa: BLE001
        cur = None
    for f in reversed(files):
        try:
            js = json.load(open(f))
        except (OSError, ValueError):
            continue
        v = js.get("_cell_launch_mean_traffic_bytes")
        if v:
            sha = js.get("_lib_src_sha16")
            same = bool(sha) and sha == cur
            return (float(v) if same else None), {
                "bytes_per_cell_launch_hbm_side": float(v), "file": "profiles/" + os.path.basename(f),
                "lib_src_sha16_of_profile": sha, "lib_src_sha16_of_this_run": cur,
                "same_sources": same,
                "note": "PMC passes of the same bench command (profiles/collect.sh), not measured "
                        "inside this run"}
    return None, None


def rocprof_launch_us(pattern="r*_bench_kernel_stats.csv"):
    """Average launch duration of the chain kernels in the newest committed rocprofv3
    --kernel-trace --stats summary of this command (profiles/*_kernel_stats.csv): the PROFILED
    figure, reported next to the unprofiled HIP-event one (the kernel-trace instrumentation adds
    ~0.9 us to these 4-us kernels: compare trace and events inside the profiled run,
    profiles/README.md)."""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", pattern)))
    if not files:
        return None
    tot, calls, per = 0.0, 0, {}
    try:
        for row in csv.DictReader(open(files[-1])):
            name = row["Name"]
            for key in ("cell_a_kernel", "cell_b_kernel", "bwd_a_kernel", "bwd_edge_kernel"):
                if key in name:
                    c, d = int(row["Calls"]), float(row["TotalDurationNs"])
                    tot += d
                    calls += c
                    a = per.setdefault(key, [0, 0.0])
                    a[0] += c
                    a[1] += d
    except (OSError, ValueError, KeyError):
        return None
    if not calls:
        return None
    return {"file": "profiles/" + os.path.basename(files[-1]), "mean_us": tot / calls * 1e-3,
            "per_kernel_us": {k: v[1] / v[0] * 1e-3 for k, v in per.items()}}


def _limit_threads(t):
    """Context manager capping the BLAS pools at t threads (no-op without threadpoolctl)."""
    import contextlib
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=int(t))
    except Exception:        # noqa: BLE001
        return contextlib.nullcontext()


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:        # noqa: BLE001
        return os.cpu_count() or 1


def cpu_baseline(F, r, K, B, frames, tied):
    """The reference's op graph (dense p.U_k, materialised Gram h.S_k, x.Wk_k, bias+relu, K layers
    per frame inside a loop over time; custom_layers.py:361-369 + enhance.py:161-204) restated in
    numpy fp32 (oracle/), timed on this host's cores on a bounded sample of the same workload
    (forward: the reference's Theano graph cannot run here, and its backward is Theano autodiff).
    The BLAS thread count is swept (64-row GEMMs oversubscribe a 128-thread pool) and the best is
    reported with its count."""
    from oracle import drnmf_oracle as O
    N = 2 * r
    P = O.synth_problem(B, 2, F, r, seed=7654)
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(400.0 if r >= 1000 else 50.0), lam1=np.float32(1.0))
    alt, labels = O.build_alt(N, K, params, () if tied else ("log_D", "log_alph"))
    t0 = time.perf_counter()
    Wk, Uk, bk, Sk = O.maps_dense(alt, labels, K, N, dtype=np.float32)
    t_maps = time.perf_counter() - t0
    fact = O.maps_factored(alt, labels, K, np.float32)
    us = O.u_scalars(alt, np.float32)

    def run_dense(Pp):
        t0 = time.perf_counter()
        O.cell_forward_dense(Pp["X"], Wk, Uk, bk, Sk, Pp["log_h0"], dtype=np.float32)
        return time.perf_counter() - t0

    def run_fact(Pp):
        t0 = time.perf_counter()
        O.cell_forward_factored(Pp["X"], fact, us, Pp["log_h0"], dtype=np.float32)
        return time.perf_counter() - t0
    max_threads = _blas_threads()
    sweep = {}
    cands = sorted(set([t for t in (4, 8, 16, 32, 64, 128) if t <= max_threads] + [max_threads]))
    best_t, best_rate = max_threads, 0.0
    for t in cands:
        with _limit_threads(t):
            run_dense(P)                              # warm
            sec = run_dense(P)
        sweep[str(t)] = B * 2 / sec
        if sweep[str(t)] > best_rate:
            best_rate, best_t = sweep[str(t)], t
    if frames <= 0:      # ~15 s at the best thread count
        frames = int(max(4, min(400, 15.0 * best_rate / B)))
    P = O.synth_problem(B, frames, F, r, seed=7654)
    with _limit_threads(best_t):
        t_dense = run_dense(P)
        t_fact = run_fact(P)
    cand_fb = [t for t in (8, 16, 32, 64) if t <= max_threads] or [max_threads]
    fb = cpu_fwd_bwd(F, r, K, B, max(2, min(64, frames // 4)), tied, cand_fb)
    host = host_description()
    fwd_sample = ("numpy fp32 restatement of the reference op graph (dense U, Gram S, per-step GEMMs), "
                  "FORWARD of the recurrent cell on %d utterances x %d frames of the same workload "
                  "(Gram/matrix build %.1f s excluded); BLAS threads swept, best of %s reported (host "
                  "exposes %d); factored form on the same sample and threads: %.0f frames/s" %
                  (B, frames, t_maps, sorted(int(k) for k in sweep), max_threads, B * frames / t_fact))
    # `value` is the metric's own step (forward + loss + backward) on the host, beside the fwd+bwd
    # headline; the forward of the reference's dense op graph is kept next to it
    return {
        "value": fb["value"], "unit": "frames/s", "cores": fb["cores"], "kind": "port",
        "threads": fb["cores"], "passes": fb["passes"], "min": fb["min"], "max": fb["max"], "spread": fb["spread"],
        "thread_sweep_frames_per_s": fb.get("thread_sweep_frames_per_s"),
        "cpu_model": host["cpu_model"], "physical_cores": host["physical_cores"],
        "hardware_threads": host["hardware_threads"], "blas": host["blas"],
        "sample": "the metric's step on the host: " + fb["sample"] + "; the same model and batch shape "
                  "as the headline (forward alone, reference op graph: `forward_only`)",
        "forward_only": {"value": B * frames / t_dense, "unit": "frames/s", "cores": int(best_t),
                         "kind": "port", "sample": fwd_sample,
                         "thread_sweep_frames_per_s": sweep},
    }


def host_description():
    """CPU model, physical cores, hardware threads and BLAS of this host (SURVEY.md 8d asks for them beside
    the CPU baseline)."""
    model, phys, logical = None, set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "processor":
                logical += 1
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    blas = []
    try:
        from threadpoolctl import threadpool_info
        blas = sorted(set("%s %s" % (p_.get("internal_api", "?"), p_.get("version", "")) for p_ in threadpool_info()))
    except Exception:        # noqa: BLE001
        pass
    try:
        import torch
        cfg = torch.__config__.show()
        tb = [n for n in ("MKL", "OpenBLAS", "BLIS", "Eigen") if ("USE_" + n.upper() + "=ON") in cfg or
              ("BLAS_INFO=" + n.lower()) in cfg]
        if tb:
            blas.append("torch: " + "/".join(tb))
    except Exception:        # noqa: BLE001
        pass
    return {"cpu_model": model, "physical_cores": len(phys) or None, "hardware_threads": logical or os.cpu_count(),
            "blas": ", ".join(blas) or None}


def cpu_fwd_bwd(F, r, K, B, frames, tied, threads, passes=5):
    """The metric's own step on the host: forward + loss + backward of the whole model by torch-CPU
    fp32 autograd of the oracle restatement (oracle/drnmf_torch_ref.py, factored form -- the
    reference gets its gradients from Theano autodiff of the same graph, enhance.py:1071-1073), on
    a bounded sample.  A MEASUREMENT, not a sample of one: the intra-op thread count is
    chosen by a sweep on an eighth of the sample, then one untimed pass and `passes` timed ones; the value is
    the MEDIAN, the spread is reported with it."""
    import torch
    from oracle import drnmf_oracle as O
    from oracle import drnmf_torch_ref as TR
    N = 2 * r
    P = O.synth_problem(B, frames, F, r, seed=7654)
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(400.0 if r >= 1000 else 50.0), lam1=np.float32(1.0))
    alt, labels = O.build_alt(N, K, params, () if tied else ("log_D", "log_alph"))
    old = torch.get_num_threads()
    try:
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        a = {k: f32(v).requires_grad_(k.startswith("log_D") or k.startswith("log_alph"))
             for k, v in alt.items()}
        lh0 = f32(P["log_h0"]).requires_grad_(True)
        kc = f32(np.log(1e-7 + P["W"][:, :r]).T).requires_grad_(True)
        kn = f32(np.log(1e-7 + P["W"][:, r:]).T).requires_grad_(True)
        x, y = f32(P["X"]), f32(P["Y"])

        def one_pass(nf):
            w = torch.ones((B, nf), dtype=torch.float32)
            t0 = time.perf_counter()
            loss, _, _ = TR.model_loss(x[:, :nf], y[:, :nf], w, a, labels, K, lh0, kc, kn)
            loss.backward()
            dt = time.perf_counter() - t0
            for t_ in list(a.values()) + [lh0, kc, kn]:
                t_.grad = None
            return dt
        sweep = {}
        if isinstance(threads, (list, tuple)):
            nf = max(2, frames // 8)
            for t in threads:
                torch.set_num_threads(int(t))
                one_pass(nf)
                sweep[int(t)] = B * nf / min(one_pass(nf), one_pass(nf))
            threads = max(sweep, key=sweep.get)
        torch.set_num_threads(int(threads))
        one_pass(frames)                              # allocator and BLAS warmed up
        dts = sorted(one_pass(frames) for _ in range(passes))
    finally:
        torch.set_num_threads(old)
    med = dts[len(dts) // 2]
    rates = [B * frames / d for d in dts]
    out = {"value": B * frames / med, "unit": "frames/s", "cores": int(threads), "kind": "port",
           "passes": passes, "min": min(rates), "max": max(rates),
           "spread": (max(rates) - min(rates)) / (B * frames / med),
           "sample": "torch-CPU fp32 autograd of the factored restatement (forward + loss + backward, "
                     "no optimiser step), %d utterances x %d frames, median of %d passes behind one untimed "
                     "pass, %d intra-op threads" % (B, frames, passes, int(threads))}
    if sweep:
        out["thread_sweep_frames_per_s"] = {str(k): v for k, v in sorted(sweep.items())}
    return out


def config1_bench(torch, dev):
    """BASELINE configs[0] -- the reference's own CPU-runnable case: W 513 x 200, K = 10, ONE
    utterance (T = 1000 synthetic frames) -- on the GPU (cell + head) and on the host (oracle port,
    same two op-graph forms as cpu_baseline).  B = 1 is where the GPU advantage is smallest: one
    16-row MFMA tile is 1/16 full and every launch is pure latency."""
    from drnmf_amd import layers, ops
    from oracle import drnmf_oracle as O
    F, r, K, T = 513, 100, 10, 1000
    N = 2 * r
    W, log_h0, X = synth_on_device(torch, dev, 1, T, F, r, seed=11)
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
             alph=50.0, lam1=1.0, params_trainable=["log_D", "log_alph"],
             params_untied=["log_D", "log_alph"])
    model = layers.build_unfolded_snmf(p, device=dev)
    model.cell.log_h0.copy_(torch.from_numpy(log_h0))
    h_buf = torch.empty((1, T, N), dtype=torch.float32, device=dev)
    m_buf = torch.empty((1, T, F), dtype=torch.float32, device=dev)

    def step():
        h = model.cell.call(X, mask_value=-1., out=h_buf)
        ops.head_forward(h, model.clean.kernel, model.noise.kernel, out=m_buf)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    out = {"shape": "F=%d N=%d K=%d B=1 T=%d untied" % (F, N, K, T),
           "gpu_frames_per_s": T / sec, "gpu_ms_per_utterance": sec * 1e3,
           "gpu_launch_us": sec / (T * (2 * K - 1)) * 1e6}
    Xh = X.cpu().numpy()
    params = dict(W=W, U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(50.0), lam1=np.float32(1.0))
    alt, labels = O.build_alt(N, K, params, ("log_D", "log_alph"))
    Wk, Uk, bk, Sk = O.maps_dense(alt, labels, K, N, dtype=np.float32)
    fact, us = O.maps_factored(alt, labels, K, np.float32), O.u_scalars(alt, np.float32)
    best = {}
    for t in (1, 4, 16):
        with _limit_threads(t):
            t0 = time.perf_counter()
            O.cell_forward_dense(Xh, Wk, Uk, bk, Sk, log_h0, dtype=np.float32)
            td = time.perf_counter() - t0
            t0 = time.perf_counter()
            hf = O.cell_forward_factored(Xh, fact, us, log_h0, dtype=np.float32)
            tf = time.perf_counter() - t0
        best[str(t)] = {"dense_graph_frames_per_s": T / td, "factored_frames_per_s": T / tf}
    out["cpu_port_by_blas_threads"] = best
    out["cpu_best_frames_per_s"] = max(max(v.values()) for v in best.values())
    hd = h_buf.cpu().numpy()
    out["max_abs_dh_vs_cpu_port_rel"] = float(np.max(np.abs(hd - hf)) / max(np.max(np.abs(hf)), 1e-30))
    del model, X, h_buf, m_buf
    torch.cuda.empty_cache()
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start N ranks with torch.distributed.run as
    a CHILD process (nothing in this process has touched the GPU yet) and leave with its code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without a launcher; starting: %s\n" % (a.gpus, " ".join(cmd)))
    return subprocess.call(cmd)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and "RANK" not in os.environ and a.gpus > 1:
            raise SystemExit(spawn_ranks(a))
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch with python -m "
                         "torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr "
                         "127.0.0.1 --master-port P bench.py --gpus %d ..." %
                         (a.gpus, world, a.gpus, a.gpus))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (test aids: DRNMF_BENCH_BACKEND=gloo and DRNMF_BENCH_DEVICE=<i> let the N > 1 path run with
        # several ranks on ONE GPU, where RCCL refuses duplicate devices; the gradient all-reduce
        # then goes through torch.distributed as well, DRNMF_DP_BACKEND=torch)
        if "DRNMF_BENCH_DEVICE" in os.environ:
            local = int(os.environ["DRNMF_BENCH_DEVICE"])
        backend = os.environ.get("DRNMF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            os.environ.setdefault("DRNMF_DP_BACKEND", "torch")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import __graft_entry__ as G
    G.build()
    from drnmf_amd import layers, ops

    B, T, F, r, K = a.batch, a.frames, a.bins, a.r, a.layers
    N = 2 * r
    W, log_h0, X, Y = synth_on_device(torch, dev, B, T, F, r, seed=7654 + rank, want_clean=True)
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
             alph=400.0 if r >= 1000 else 50.0, lam1=1.0, params_trainable=["log_D", "log_alph"])
    if not a.tied:
        p["params_untied"] = ["log_D", "log_alph"]
    model = layers.build_unfolded_snmf(p, device=dev)
    model.cell.log_h0.copy_(torch.from_numpy(log_h0))
    wts = torch.ones((B, T), dtype=torch.float32, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(step, steps, warmup):
        """W untimed + K timed steps, barrier + synchronize on both sides, max over ranks; returns
        (wall seconds, HIP-event milliseconds of this rank's stream)."""
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall = float(tt.item())
        return wall, e0.elapsed_time(e1)

    n_chain = T * (2 * K - 1)            # launches of one pass over the batch (forward or BPTT)
    flops_per_launch = 2.0 * B * F * N    # one B x F x N contraction
    traffic, traffic_src = pmc_traffic()
    prof = rocprof_launch_us()

    # ---------------- headline: forward + loss + BPTT + Adam (the metric's "fwd+bwd") ----------
    out = {}
    train = None
    collective = None
    comm_seen = None
    if not a.forward_only:
        def on_timeout():
            # a collective that never comes back: say so and leave with a failure code (the
            # launcher must see it); never re-exec a process that has touched the GPU
            if rank == 0:
                print(json.dumps({"metric": "STFT frames/sec (fwd+bwd)", "value": None,
                                  "error": "training step timed out (collective?)",
                                  "n_gpus": world}), flush=True)
            os._exit(3)

        if world > 1:
            # the gradient all-reduce goes through libdrnmf's RCCL communicator (C ABI).  A failure
            # to create it ENDS the run with a non-zero code on every rank: a line whose collective
            # is not the product's must never be mistaken for the N-GPU number.  Only an explicit
            # DRNMF_DP_BACKEND=torch (several test ranks on one GPU) selects torch.distributed.
            from drnmf_amd import dp
            if os.environ.get("DRNMF_DP_BACKEND", "rccl") == "torch":
                collective = "torch.distributed all-reduce (DRNMF_DP_BACKEND=torch, set explicitly)"
            else:
                def init_comm():
                    dp.comm_init(dev)
                    return {}
                res = run_guarded(init_comm, 300.0, on_timeout)
                failed = torch.tensor([1.0 if "error" in res else 0.0], device=dev)
                dist.all_reduce(failed, op=dist.ReduceOp.MAX)        # every rank learns of it
                if float(failed.item()) > 0 and os.environ.get("DRNMF_BENCH_STRICT_COMM") == "1":
                    if rank == 0:
                        print(json.dumps({"metric": "STFT frames/sec (fwd+bwd)", "value": None,
                                          "n_gpus": world,
                                          "error": "drnmf_comm_init failed (%s); DRNMF_BENCH_STRICT_COMM=1: "
                                                   "no fallback" %
                                                   res.get("error", "on another rank")}), flush=True)
                    raise SystemExit(5)
                if float(failed.item()) > 0:
                    # The library's own communicator could not be created (it has never met more than one GPU:
                    # DESIGN.md section 5).  The SAME collective -- one all-reduce(sum) of the flat buffer per
                    # step, over RCCL -- is then issued through torch.distributed's communicator (backend
                    # "nccl" IS RCCL on ROCm) on EVERY rank, and the line says so where it cannot be missed.
                    why = res.get("error", "failed on another rank")
                    sys.stderr.write("bench.py[rank %d]: drnmf_comm_init FAILED (%s); the gradient all-reduce "
                                     "goes through torch.distributed (RCCL) instead\n" % (rank, why))
                    os.environ["DRNMF_DP_BACKEND"] = "torch"
                    try:
                        dp.comm_destroy()
                    except Exception:        # noqa: BLE001 -- nothing to destroy on the rank that failed
                        pass
                    collective = ("torch.distributed all-reduce (backend %s%s) -- FALLBACK: drnmf_comm_init "
                                  "failed (%s)" % (dist.get_backend(),
                                                   " = RCCL" if dist.get_backend() == "nccl" else "", why[:200]))
                else:
                    collective = "drnmf_allreduce_grads (RCCL communicator owned by the library handle)"
                    comm_seen = dp.comm_info(dev)          # (rank, world) as RCCL's communicator has them
                    if comm_seen is None or comm_seen[1] != world:
                        raise SystemExit("bench.py: the library's communicator reports %r, expected "
                                         "world %d" % (comm_seen, world))

        def headline():
            model.compile(lr=1e-3)
            model.phase_events = {}
            losses = []
            wall, ev_ms = timed(lambda: losses.append(model.train_on_batch(X, Y, wts)),
                                a.steps, a.warmup)
            pe = model.phase_events            # events of the LAST timed step
            phases = {k: v[0].elapsed_time(v[1]) for k, v in pe.items()}
            model.phase_events = None
            # one more step with the backward's phase boundaries bracketed (synchronising aid)
            model.backward_profile = {}
            model.train_on_batch(X, Y, wts)
            bp = dict(model.backward_profile)
            model.backward_profile = None
            ar = None
            if world > 1:      # the step's one collective alone: all-reduce of the flat buffer, 5 times
                from drnmf_amd import dp as _dp
                flat = model._flat
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                _dp.allreduce_sum_(flat)
                torch.cuda.synchronize()
                barrier()
                e0.record()
                for _ in range(5):
                    _dp.allreduce_sum_(flat)
                e1.record()
                torch.cuda.synchronize()
                ar = {"bytes": int(flat.numel()) * 4, "ms": e0.elapsed_time(e1) / 5.0}
            x3 = None
            if world == 1 and not a.no_extras:
                # the SAME step with the time-batched products (weight gradients, mask head and its backward) in
                # the split-operand mode: a second, separately labelled figure -- never `value`
                prev = ops.set_matrix_mode("bf16x3")
                try:
                    model.phase_events = {}
                    xl = []
                    xwall, xev = timed(lambda: xl.append(model.train_on_batch(X, Y, wts)), max(2, min(a.steps, 5)), 1)
                    model.phase_events = None
                    model.backward_profile = {}
                    model.train_on_batch(X, Y, wts)
                    xbp = dict(model.backward_profile)
                    model.backward_profile = None
                    x3 = dict(wall=xwall, steps=max(2, min(a.steps, 5)), bp=xbp, loss_last=float(xl[-1]))
                finally:
                    ops.set_matrix_mode(prev)
            return dict(wall=wall, ev_ms=ev_ms, phases=phases, bp=bp, losses=losses, allreduce=ar, x3=x3)
        train = run_guarded(headline, 900.0, on_timeout)
    if train is not None and "error" not in train:
        wall, ev_ms = train["wall"], train["ev_ms"]
        value = world * B * T * a.steps / wall
        fwd_ms, bwd_chain_ms = train["phases"]["cell_forward"], train["bp"]["chain_ms"]
        launches = n_chain + train["bp"]["chain_launches"]
        launch_us = (fwd_ms + bwd_chain_ms) * 1e3 / launches
        ach = flops_per_launch / (launch_us * 1e-6) / 1e12
        flops_step = (12.0 * F * N * K - 2.0 * F * N) * B * T
        out = {
            "metric": "STFT frames/sec (fwd+bwd), %d-bin x %d-frame, K=%d unrolls" % (F, T, K),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": wall * 1e3 / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] shape, forward + loss + BPTT + Adam "
                                   "(train_on_batch): K=%d unrolled SNMF, W %dx%d, %s "
                                   "log_D/log_alph, batch %d x %d frames per GPU%s" %
                                   (K, F, N, "tied" if a.tied else "untied", B, T,
                                    ", one RCCL all-reduce of the flat gradient per step"
                                    if world > 1 else ""),
                       "B_per_gpu": B, "T": T, "F": F, "N": N, "K": K, "untied": not a.tied,
                       "collective": collective,
                       "comm_rank_world_as_reported_by_drnmf_comm_info": comm_seen,
                       "allreduce_bytes": (train.get("allreduce") or {}).get("bytes"),
                       "allreduce_ms": (train.get("allreduce") or {}).get("ms")},
            "roofline": {
                "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_from_profile": traffic_src,
                "kernel": "the chain kernels cell_a / cell_b (forward) and cell_b / bwd_a (BPTT): "
                          "each launch is one B x F x N contraction",
                "launch_us": launch_us, "flops_per_launch": flops_per_launch,
                "launches_per_step": launches,
                "duration_source": "HIP events on the launch stream, unprofiled run: (cell forward "
                                   "of the last timed step + sequential pass of the BPTT bracketed "
                                   "by drnmf_cell_backward_profile) / launches; includes the "
                                   "~1.5 us launch boundary, as a zero-gap kernel trace does",
                "launch_us_rocprof": prof,
                "frac_rocprof": (flops_per_launch / (prof["mean_us"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS)
                if prof else None,
                "frac_rocprof_source": (prof or {}).get("file"),
                "algorithmic_bytes_per_launch": 4.0 * (F * N + B * (N + F)),
                "algorithmic_bytes_note": "one layer's dictionary (one packing, F x N fp32) + the activations a "
                                          "launch exchanges (h in, residual out or the reverse: B (N + F) fp32); "
                                          "`traffic` is the PMC figure of the same launch (L2-side, every XCD's "
                                          "fetch of a shared operand counted)",
            },
            "step_breakdown_ms": {
                "cell_forward_chain": fwd_ms, "head_and_loss": train["phases"]["head_and_loss"],
                "cell_backward_total": train["phases"]["cell_backward"],
                "bptt_sequential_pass": bwd_chain_ms,
                "bptt_time_batched_weight_gradients": train["bp"]["batched_ms"],
                "hip_event_ms_per_step": ev_ms / a.steps},
            "whole_step_tflops_algorithmic": flops_step / (ev_ms / a.steps * 1e-3) / 1e12,
            "whole_step_frac_of_f32_mfma_peak":
                flops_step / (ev_ms / a.steps * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "loss_first": float(train["losses"][0]), "loss_last": float(train["losses"][-1]),
        }
        if train.get("x3"):
            x3 = train["x3"]
            out["headline_bf16x3"] = {
                "value": world * B * T * x3["steps"] / x3["wall"], "unit": "frames/s",
                "ms_per_step": x3["wall"] * 1e3 / x3["steps"],
                "dtype": "f32 via bf16x3 operands, f32 accumulate (the time-batched products: weight gradients, "
                         "mask head and its backward; the recurrent chain kernels contract in exact f32 in both)",
                "bptt_time_batched_weight_gradients_ms": x3["bp"]["batched_ms"],
                "bptt_sequential_pass_ms": x3["bp"]["chain_ms"],
                "loss_last": x3["loss_last"],
                "note": "a second, separately labelled line (VERDICT r5 item 1): `value` above is the native-f32 "
                        "figure; same model, batch and step"}
    elif train is not None:
        out = {"metric": "STFT frames/sec (fwd+bwd), %d-bin x %d-frame, K=%d unrolls" % (F, T, K),
               "value": None, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "error": train["error"]}

    # release the training state (hall, dz, dR: ~60 GB at the headline shape) before the rest
    model._opt_state = model._flat = model._gview = model._mflat = model._vflat = None
    model._adam_table = None
    model.cell._ws.clear()
    del Y
    torch.cuda.empty_cache()

    # ---------------- forward alone: recurrent cell + mask head (north-star target) ------------
    if a.operand_f16:
        p2 = dict(p, operand_dtype="float16")
        model = layers.build_unfolded_snmf(p2, device=dev)
        model.cell.log_h0.copy_(torch.from_numpy(log_h0))
    h_buf = torch.empty((B, T, N), dtype=torch.float32, device=dev)
    m_buf = torch.empty((B, T, F), dtype=torch.float32, device=dev)
    cell_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cell_ms = []

    def fwd_step():
        cell_ev[0].record()
        h = model.cell.call(X, mask_value=-1., out=h_buf)
        cell_ev[1].record()
        ops.head_forward(h, model.clean.kernel, model.noise.kernel, out=m_buf)
        cell_ev[1].synchronize()
        cell_ms.append(cell_ev[0].elapsed_time(cell_ev[1]))
    fwall, fev_ms = timed(fwd_step, a.steps, a.warmup)
    cell_ms = cell_ms[-a.steps:]
    ok = bool(torch.isfinite(m_buf).all().item()) and float(m_buf.min()) > 0.0
    f_launch_us = (sum(cell_ms) / len(cell_ms)) * 1e3 / n_chain
    f_ach = flops_per_launch / (f_launch_us * 1e-6) / 1e12
    whole = (B * T * a.steps) * 4.0 * F * N * K / (fev_ms * 1e-3) / 1e12
    per_kernel = None
    if rank == 0 and world == 1:
        try:   # launch-by-launch HIP events over the first frames (plain launches, no graph)
            desc = model.cell._desc(B, T)
            per_kernel = ops.cell_profile(X, -1., model.cell._params_block, desc, model.cell.log_h0,
                                          model.cell._u, h_buf, model.cell._ws[(B, T)], frames=8)
        except Exception as e:        # noqa: BLE001
            per_kernel = {"error": repr(e)[:200]}
    forward = {
        "value": world * B * T * a.steps / fwall, "unit": "frames/s",
        "ms_per_step": fwall * 1e3 / a.steps,
        "dtype": "f16 operands, f32 accumulate" if a.operand_f16 else "f32",
        "workload": "BASELINE configs[1]: K=%d unrolled SNMF forward (recurrent cell + mask head), "
                    "same model and batch" % K,
        "roofline": {"bound": "mfma", "achieved": f_ach, "peak": PEAK_F32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": f_ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                     "traffic_from_profile": traffic_src,
                     "kernel": "cell_a_kernel / cell_b_kernel (mean over the 2K-1 launches of a frame)",
                     "launch_us": f_launch_us, "flops_per_launch": flops_per_launch,
                     "duration_source": "HIP events around the cell's graph replays, unprofiled "
                                        "run, / T(2K-1) launches",
                     "launch_us_rocprof": prof,
                     "algorithmic_bytes_per_launch": 4.0 * (F * N + B * (N + F)),
                     "per_kernel_us_plain_launches_with_events": per_kernel},
        "whole_forward_tflops": whole,
        "whole_forward_frac_of_f32_mfma_peak": whole / PEAK_F32_MFMA_TFLOPS,
        "finite_positive_masks": ok,
    }
    if a.forward_only or not out:
        out = {"metric": "STFT frames/sec (fwd only: --forward-only), %d-bin x %d-frame, K=%d "
                         "unrolls" % (F, T, K),
               "value": forward["value"], "unit": "frames/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": forward["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": forward["dtype"], "data": "synthetic",
               "config": {"workload": forward["workload"], "B_per_gpu": B, "T": T, "F": F, "N": N,
                          "K": K, "untied": not a.tied},
               "roofline": forward["roofline"]}
    out["forward"] = forward

    extra = {}
    if rank == 0 and world == 1 and not a.no_extras:
        def safe(fn, *args, **kw):
            """An extra line must never cost the headline line."""
            try:
                return fn(*args, **kw)
            except Exception as e:       # noqa: BLE001 -- reported in the JSON line instead
                torch.cuda.empty_cache()
                return {"error": repr(e)[:300]}
        del h_buf, m_buf
        torch.cuda.empty_cache()
        if not a.no_ista:
            extra["ista_frame_parallel"] = safe(both_modes, ista_bench, lambda d: d["gemm_launch_us"],
                                                torch, dev, F, N, K, W)
            extra["mu_inference"] = safe(both_modes, mu_bench, lambda d: d["ms_per_iteration"], torch, dev, F, N, W)
            extra["stft_front_end"] = safe(stft_bench, torch, dev)
            extra["dictionary_training"] = safe(both_modes, snmf_train_bench,
                                                lambda d: d["ed"]["ms_per_iteration"], torch, dev, F, r)
            dt = extra["dictionary_training"]
            if "bf16x3" in dt and "kl" in dt and "kl" in dt["bf16x3"]:
                dt["speedup_bf16x3_kl"] = dt["kl"]["ms_per_iteration"] / dt["bf16x3"]["kl"]["ms_per_iteration"]
        extra["config1_single_utterance"] = safe(config1_bench, torch, dev)
        if not a.no_slab:
            extra["inference_slab_250"] = safe(slab_bench, torch, dev, F, r, K, T, host_slabs=3)
            # the same forward (cell + mask head) over batch sizes: from 128 rows on the batch runs as
            # independent sub-batches on side streams (csrc/cell_shared.h Workspace::split)
            sweep = []
            for b_sw in (64, 128, 250, 512, 1024):
                res = safe(slab_bench, torch, dev, F, r, K, min(T, 400), slab=b_sw)
                sweep.append({"B": b_sw, "frames": min(T, 400),
                              **({"error": res["error"]} if "error" in res else
                                 {"frames_per_s": res["frames_per_s"],
                                  "frac_of_f32_mfma_peak": res["frac_of_f32_mfma_peak"]})})
            extra["inference_batch_sweep"] = sweep
            # the same slab for the shipped small model (params_unfolded_snmf_*.yaml: N_fft = 512, r = 100,
            # K = 5): 16 persistent row-tile chains, two per XCD (csrc/cell_gram_persist.h)
            extra["inference_slab_250_shipped_r100"] = safe(slab_bench, torch, dev, 257, 100, 5, 500)
            extra["inference_ragged_dataset"] = safe(ragged_inference_bench, torch, dev)
            # ... and with the headline's dictionary (F = 513, N = 2000, K = 25): there the GPU, not the host's
            # staging copies, bounds `predict`
            extra["inference_ragged_dataset_configs1_model"] = safe(ragged_inference_bench, torch, dev, F=F, r=r, K=K,
                                                                    T=T, n=1000, slab=250)
            extra["reference_op_graph_dense_kernel"] = safe(dense_graph_bench, torch, dev, F, r, K, B)
        if not a.no_config5:
            extra["config5_shape"] = safe(config5_bench, torch, dev)
        if not a.no_train:
            del X
            torch.cuda.empty_cache()
            extra["train_step_configs2"] = safe(both_modes, train_bench, lambda d: d["ms_per_step"], torch, dev)
            # the other shipped dictionary size (params_unfolded_snmf_ea1e7d48: r = 100, K = 5)
            extra["train_step_configs2_r100"] = safe(both_modes, train_bench, lambda d: d["ms_per_step"], torch, dev,
                                                     shape=(32, 500, 257, 100, 5))
    if world > 1 and not a.no_extras:
        # ---- the 8-GPU configurations of BASELINE.json (configs[3], configs[4]): every rank takes part ----
        # (DRNMF_BENCH_TINY=1: the same control flow at toy shapes -- several ranks on ONE GPU over gloo,
        # profiles/r05_n8_gloo_one_gpu.json; never a measurement)
        tiny = os.environ.get("DRNMF_BENCH_TINY") == "1"
        del h_buf, m_buf, X
        model = None
        torch.cuda.empty_cache()

        def extras_timeout():
            # a collective that never comes back inside an EXTRA must not cost the headline line
            if rank == 0:
                out["extra"] = dict(extra, error="a multi-rank extra timed out")
                print(json.dumps(out), flush=True)
            os._exit(0 if out.get("value") is not None else 4)

        def multi():
            res = {}
            if not a.no_train:
                dpt = {}
                for r_dp in (100, 1000):
                    shp = (4, 12, 33, 8 if r_dp == 100 else 24, 2) if tiny else (32, 500, 257, r_dp, 5)
                    dpt["r%d" % r_dp] = dp_train_bench(torch, dev, dist, world, rank, shp,
                                                       steps=2 if tiny else 10, warmup=1 if tiny else 3)
                res["configs3_dp_training"] = dpt
            if not a.no_config5:
                res["configs4_c5"] = c5_replicas_bench(torch, dev, dist, world, rank,
                                                       frames=4 if tiny else 16, B=4 if tiny else 64,
                                                       shape=(65, 40, 3) if tiny else (1025, 4000, 50))
            return res
        res = run_guarded(multi, 1200.0, extras_timeout)
        # (an exception on one rank only would leave the others inside a collective: every rank learns)
        bad = torch.tensor([1.0 if "error" in res else 0.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        extra.update(res if "error" not in res else {"multi_rank_extras_error": res["error"]})
        if float(bad.item()) > 0 and "error" not in res:
            extra["multi_rank_extras_error"] = "failed on another rank"
    out["extra"] = extra
    if rank == 0 and world > 1 and not a.no_cpu_baseline:
        # (the other ranks wait in destroy_process_group below; the headline's timed region is long over)
        try:
            out["cpu_baseline"] = cpu_baseline(F, r, K, B, a.cpu_frames, a.tied)
        except Exception as e:           # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)[:300]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(F, r, K, B, a.cpu_frames, a.tied)
        except Exception as e:           # noqa: BLE001 -- the GPU line is still valid without it
            out["cpu_baseline"] = {"error": repr(e)[:300]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        from drnmf_amd import dp
        dp.comm_destroy()
        dist.destroy_process_group()
    if out.get("value") is None:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
