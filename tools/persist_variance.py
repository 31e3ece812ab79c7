#!/usr/bin/env python3
"""Run-to-run spread of the persistent chains' forward: usage persist_variance.py B T F r K [reps]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import layers
import bench
B, T, F, r, K = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
dev = torch.device('cuda:0')
N = 2 * r
W, log_h0, X = bench.synth_on_device(torch, dev, B, T, F, r, seed=1)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
         alph=0.4 * r, lam1=1.0, params_trainable=["log_D", "log_alph"], params_untied=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
out = torch.empty((B, T, N), device=dev)
ms = []
for i in range(reps + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); model.cell.call(X, mask_value=-1., out=out); e1.record(); torch.cuda.synchronize()
    if i >= 2:
        ms.append(e0.elapsed_time(e1))
ms = np.array(ms)
print("first 8:", np.round(ms[:8], 3)); print("B=%d T=%d F=%d N=%d K=%d: min %.3f median %.3f max %.3f ms; sorted tail %s" %
      (B, T, F, N, K, ms.min(), np.median(ms), ms.max(), np.round(np.sort(ms)[-4:], 3)))
