"""Is the split-operand mode's error against fp64 biased?  (ISTA K = 1: one H W^T and one R W product.)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(5)
n, F, N = 2048, 513, 2000
W = torch.rand((F, N), generator=g, device=dev) ** 4
W = W / (W * W).sum(0, keepdim=True).sqrt()
Ht = (torch.rand((n, N), generator=g, device=dev) < 0.05) * torch.rand((n, N), generator=g, device=dev) * 5.0
X = Ht @ W.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
for h0 in (0.1, 0.125, 'rand'):
    H0 = torch.rand((n, N), generator=g, device=dev) * 0.2 if h0 == 'rand' else torch.full((n, N), h0, device=dev)
    Xd, Wd = X.double(), W.double()
    R64 = Xd - H0.double() @ Wd.t()
    G64 = R64 @ Wd
    ref = torch.clamp(H0.double() + G64 / 400.0 - 1.0 / 400.0, min=0.0)
    for mode in ('f32', 'bf16x3'):
        ops.set_matrix_mode(mode)
        H = ops.ista_forward(X, W, H0.clone(), 1.0, 400.0, 1)
        torch.cuda.synchronize()
        d = (H.double() - ref)[ref > 0]
        s = ref.abs().max().item()
        print('H0=%s %s: max %.2e rms %.2e mean %+.2e (of max |H| %.3g); mean/rms %+.2f' % (h0, mode, d.abs().max().item() / s, d.pow(2).mean().sqrt().item() / s, d.mean().item() / s, s, d.mean().item() / d.pow(2).mean().sqrt().item()))
ops.set_matrix_mode('f32')
