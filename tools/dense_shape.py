"""Time of the general dense-matrix cell (csrc/cell_dense.hip) at an arbitrary shape (measurement aid):
    python tools/dense_shape.py B T F N K
"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
B, T, F, N, K = [int(v) for v in sys.argv[1:6]]
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
U = torch.randn((K, N, N), generator=g, device=dev) * (0.5 / N ** 0.5)
S = torch.randn((K - 1, N, N), generator=g, device=dev) * (0.5 / N ** 0.5)
W = torch.randn((K, F, N), generator=g, device=dev) * (0.5 / F ** 0.5)
b = torch.zeros((K, N), device=dev)
X = torch.rand((B, T, F), generator=g, device=dev)
h0 = torch.zeros(N, device=dev)
desc = ops.make_dense_desc(B, T, F, N, K)
P = ops.dense_prepare_params(desc, U, S if K > 1 else None, W, b)
ws = ops.dense_workspace(desc, dev)
out = torch.empty((B, T, N), device=dev)
ops.dense_cell_forward(X, None, P, desc, h0, out=out, workspace=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.dense_cell_forward(X, None, P, desc, h0, out=out, workspace=ws); e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
fl = 2.0 * B * T * N * (K * (N + F) + (K - 1) * N)
print(sys.argv[1:], '%.2f ms, %.1f us per layer-step, %.1f k frames/s, %.1f TFLOP/s executed' %
      (ms, ms * 1e3 / (T * K), B * T / ms, fl / ms / 1e9))
