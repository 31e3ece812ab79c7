"""Would running the frame-parallel iterations chunk by chunk (frames are independent: a chunk's H / X / residual
stay in the Infinity Cache across the K iterations) beat one pass over all frames per iteration?  Emulated at the
Python level: ops.ista_forward / ops.mu_forward on row slices."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
n, F, N, K = 32768, 513, 2000, 25
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(1)
W = torch.rand((F, N), generator=g, device=dev) ** 4
W = W / (W * W).sum(0, keepdim=True).sqrt()
Ht = (torch.rand((n, N), generator=g, device=dev) < 0.02) * torch.rand((n, N), generator=g, device=dev) * 5.0
X = Ht @ W.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


for mode in ('f32', 'bf16x3'):
    ops.set_matrix_mode(mode)
    for chunk in (32768, 16384, 8192, 4096):
        H = torch.full((n, N), 0.1, device=dev)
        def run():
            for c0 in range(0, n, chunk):
                ops.ista_forward(X[c0:c0 + chunk], W, H[c0:c0 + chunk], 1.0, 400.0, K)
        sec = timed(run)
        print('ISTA %s chunk %5d: %.2f ms, %.1f TFLOP/s-eq' % (mode, chunk, sec * 1e3, n * 4.0 * F * N * K / sec / 1e12), flush=True)
    for chunk in (32768, 16384, 8192):
        H = torch.rand((n, N), generator=g, device=dev)
        def run():
            for c0 in range(0, n, chunk):
                ops.mu_forward(X[c0:c0 + chunk], W, H[c0:c0 + chunk], 5.0, 20, beta=2.0)
        sec = timed(run)
        print('MU   %s chunk %5d: %.3f ms per iteration' % (mode, chunk, sec * 1e3 / 20), flush=True)
ops.set_matrix_mode('f32')
