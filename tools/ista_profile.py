"""Frame-parallel ISTA-ED at an arbitrary shape (for rocprofv3 --kernel-trace --stats):
    python tools/ista_profile.py n F N K [f32|bf16x3]
"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
n, F, N, K = [int(v) for v in sys.argv[1:5]]
mode = sys.argv[5] if len(sys.argv) > 5 else 'f32'
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
W = torch.rand((F, N), generator=g, device=dev) ** 4
W = W / (W * W).sum(0, keepdim=True).sqrt()
Ht = (torch.rand((n, N), generator=g, device=dev) < 0.02) * torch.rand((n, N), generator=g, device=dev) * 5.0
X = Ht @ W.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
H = torch.full((n, N), 0.1, device=dev)
ops.set_matrix_mode(mode)
ops.ista_forward(X, W, H, 1.0, 400.0, 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.ista_forward(X, W, H, 1.0, 400.0, K); e1.record(); torch.cuda.synchronize()
sec = e0.elapsed_time(e1) * 1e-3
print(sys.argv[1:], '%.2f ms, %.1f TFLOP/s' % (sec * 1e3, n * 4.0 * F * N * K / sec / 1e12))
