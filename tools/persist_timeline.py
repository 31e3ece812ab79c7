#!/usr/bin/env python3
"""In-kernel timeline of gram_persist_kernel (DRNMF_TIMELINE=1 build AND run): s_memtime segment means of
wave 0 of workgroup (chain 0, tile 0) over the phases of the last launch.
usage: DRNMF_TIMELINE=1 python tools/persist_timeline.py [B T F r K]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as G
G.build()
from drnmf_amd import _capi, layers, ops
import bench

B, T, F, r, K = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (1, 1000, 513, 100, 10)
dev = torch.device("cuda", 0)
N = 2 * r
W, log_h0, X = bench.synth_on_device(torch, dev, B, T, F, r, seed=11)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
         alph=50.0, lam1=1.0, params_trainable=["log_D", "log_alph"], params_untied=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
h = model.cell.call(X, mask_value=-1.)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
h = model.cell.call(X, mask_value=-1.)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print("cell forward %.3f ms = %.3f us per phase (T*(K-1) phases)" % (ms, ms * 1e3 / (T * (K - 1))))
buf = (C.c_ulonglong * 32)()
L = _capi.lib()
L.drnmf_debug_persist_timeline.restype = C.c_int32
L.drnmf_debug_persist_timeline.argtypes = [C.c_void_p, C.c_size_t]
assert L.drnmf_debug_persist_timeline(buf, 256) == 0
v = np.array(buf[:16], dtype=np.float64)
nph = v[15]
names = ["exchanged loads issued", "operands arrived (wave 0)", "(row sums) MFMAs + LDS write",
         "workgroup barrier (slowest wave)", "reduce + update + stores issued", "stores acknowledged",
         "workgroup barrier (1)", "prefetch issued", "arrive + poll (barrier 2)"]
print("last launch: %d phases; us per phase at 2.4 GHz ticks" % nph)
for i, nme in enumerate(names):
    print("  %-36s %.3f" % (nme, v[i] / nph / 2400.0))
print("  %-36s %.3f" % ("sum", v[:9].sum() / nph / 2400.0))
