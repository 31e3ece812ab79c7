"""Measurement aid (round 4): does splitting a LARGE batch into independent sub-batches on separate
HIP streams hide the kernel boundaries of the launch-per-phase chain?

    python tools/two_stream_probe.py B T [F r K]        # e.g. 256 200

Batch rows never interact (custom_layers.py:337-338, 346-348).  Round 1 measured concurrent chains at
B = 64 (0.78x with two): there every launch is latency and a half-batch launch costs what the full one
does.  At B >= 256 a launch is mostly work, so two half-batch chains could fill each other's gaps.
Prints frames/s of: one stream with the whole batch; S = 2, 4 streams with B/S rows each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

G.build()
import bench as bm  # noqa: E402
from drnmf_amd import layers, ops  # noqa: E402

dev = torch.device('cuda:0')
B, T = int(sys.argv[1]), int(sys.argv[2])
F, r, K = (int(v) for v in sys.argv[3:6]) if len(sys.argv) >= 6 else (513, 1000, 25)
N = 2 * r
W, log_h0, X = bm.synth_on_device(torch, dev, B, T, F, r, seed=1)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
         alph=0.4 * r, lam1=1.0, params_trainable=["log_D", "log_alph"],
         params_untied=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
cell = model.cell
cell.prepare(B, T)


def run(S, reps=3):
    Bs = B // S
    desc = cell._desc(Bs, T)
    xs = [X[i * Bs:(i + 1) * Bs].contiguous() for i in range(S)]
    outs = [torch.empty((Bs, T, N), device=dev) for _ in range(S)]
    wss = [ops.cell_workspace(desc, dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    main = torch.cuda.current_stream()
    best = 1e30
    for rep in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for i in range(S):
            streams[i].wait_event(e0)
            with torch.cuda.stream(streams[i]):
                ops.cell_forward(xs[i], -1.0, cell._params_block, desc, cell.log_h0, cell._u,
                                 out=outs[i], workspace=wss[i])
        for i in range(S):
            main.wait_stream(streams[i])
        e1.record(main)
        torch.cuda.synchronize()
        if rep > 0:
            best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best, torch.cat(outs, 0)


ref = None
for S in (1, 2, 4):
    if B % (16 * S):
        continue
    sec, out = run(S)
    if ref is None:
        ref = out
    same = bool(torch.equal(out, ref))
    tf = B * T * 4.0 * F * N * K / sec / 1e12
    print('B=%d T=%d streams=%d (rows per stream %d): %.1f ms, %.0f frames/s, %.1f%% of fp32-MFMA peak, '
          'bit-identical to one stream: %s' % (B, T, S, B // S, sec * 1e3, B * T / sec, tf / 157.3 * 100, same),
          flush=True)
    del out
