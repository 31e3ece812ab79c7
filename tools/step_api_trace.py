"""HIP API calls of steady-state optimiser steps (for `rocprofv3 --hip-trace --stats`): VERDICT r3 item 6 --
no stream synchronisation and no device-to-host copy inside train_on_batch.
    python tools/step_api_trace.py B T F r K steps
Runs 3 warm-up steps, synchronises, then `steps` steps WITHOUT reading a loss, and prints the number of
synchronising HIP calls the host made in between (counted by wrapping torch.cuda.synchronize /
Tensor.tolist / Tensor.item / Tensor.cpu at the Python level; the rocprofv3 trace of the same command
is the authoritative count: run it with steps = 0 and steps = 20 and subtract the per-API call counts --
tools/hip_api_delta.py, profiles/r04g_step_hip_api_delta.txt)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
import bench as bm
from drnmf_amd import layers
B, T, F, r, K, steps = [int(v) for v in sys.argv[1:7]]
dev = torch.device('cuda:0')
N = 2 * r
W, log_h0, x, y = bm.synth_on_device(torch, dev, B, T, F, r, seed=3, want_clean=True)
w = torch.ones((B, T), dtype=torch.float32, device=dev)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
         alph=400.0 if r >= 1000 else 50.0, lam1=1.0, params_untied=["log_D", "log_alph"],
         params_trainable=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
model.compile(lr=1e-3)
for _ in range(3):
    float(model.train_on_batch(x, y, w))
torch.cuda.synchronize()
counts = {"synchronize": 0, "tolist": 0, "item": 0, "cpu": 0}
orig = {"tolist": torch.Tensor.tolist, "item": torch.Tensor.item, "cpu": torch.Tensor.cpu}
def wrap(name):
    f = orig[name]
    def g(self, *a, **k):
        if self.is_cuda:
            counts[name] += 1
        return f(self, *a, **k)
    return g
for n in orig:
    setattr(torch.Tensor, n, wrap(n))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print("MARK steady-state steps begin", flush=True)
e0.record()
losses = [model.train_on_batch(x, y, w) for _ in range(steps)]
e1.record()
print("MARK steady-state steps enqueued", flush=True)
for n in orig:
    setattr(torch.Tensor, n, orig[n])
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / max(steps, 1)
print("shape", sys.argv[1:6], "%d steps: %.3f ms per step; host-side synchronising tensor reads inside the steps: %s;"
      " losses %s" % (steps, ms, counts, ("%.6g -> %.6g" % (float(losses[0]), float(losses[-1]))) if losses else "-"))
