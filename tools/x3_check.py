"""Split-operand (bf16x3) matrix mode against exact-fp32 MFMA: error vs an fp64 reference and speed.
    python tools/x3_check.py [quick]
Frame-parallel ISTA (enhance.py:402-456) through drnmf_ista_forward in both modes."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
dev = torch.device('cuda:0')


def problem(n, F, N, seed=1):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    W = torch.rand((F, N), generator=g, device=dev) ** 4
    W = W / (W * W).sum(0, keepdim=True).sqrt()
    Ht = (torch.rand((n, N), generator=g, device=dev) < 0.02) * torch.rand((n, N), generator=g, device=dev) * 5.0
    X = Ht @ W.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
    return X, W


def ref64(X, W, H0, lam, alph, K, div):
    X, W, H = X.double(), W.double(), H0.double()
    for _ in range(K):
        Xh = H @ W.t()
        if div == 'ed':
            R = X - Xh
        else:
            R = X / Xh - 1.0
        H = torch.clamp(H + (R @ W) / alph - lam / alph, min=0.0)
    return H


def errs(H, Href):
    d = (H.double() - Href)
    s = Href.abs().max().item()
    return d.abs().max().item() / s, d.pow(2).mean().sqrt().item() / s


rows = []
for (n, F, N, K, div, alph) in [(4096, 513, 2000, 1, 'ed', 400.0), (4096, 513, 2000, 25, 'ed', 400.0),
                                (3000, 257, 200, 10, 'ed', 50.0), (2048, 513, 2000, 10, 'kl', 4000.0),
                                (1024, 1025, 8000, 5, 'ed', 1600.0), (777, 100, 36, 7, 'ed', 20.0)]:
    X, W = problem(n, F, N)
    H0 = torch.full((n, N), 0.1, device=dev)
    Hr = ref64(X, W, H0, 1.0, alph, K, div)
    out = {}
    for mode in ('f32', 'bf16x3'):
        ops.set_matrix_mode(mode)
        H = H0.clone()
        ops.ista_forward(X, W, H, 1.0, alph, K, divergence=div)
        torch.cuda.synchronize()
        out[mode] = errs(H, Hr)
    ops.set_matrix_mode('f32')
    rows.append(dict(n=n, F=F, N=N, K=K, div=div, f32_max=out['f32'][0], f32_rms=out['f32'][1],
                     x3_max=out['bf16x3'][0], x3_rms=out['bf16x3'][1]))
    print(json.dumps(rows[-1]), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    sys.exit(0)
for (n, F, N, K) in [(32768, 513, 2000, 25), (32768, 257, 2000, 25), (65536, 513, 1000, 25)]:
    X, W = problem(n, F, N)
    for mode in ('f32', 'bf16x3', 'f32', 'bf16x3'):
        ops.set_matrix_mode(mode)
        H = torch.full((n, N), 0.1, device=dev)
        ops.ista_forward(X, W, H, 1.0, 400.0, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.ista_forward(X, W, H, 1.0, 400.0, K); e1.record(); torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3
        print('%s n=%d F=%d N=%d K=%d: %.2f ms, %.1f TFLOP/s-equivalent' % (mode, n, F, N, K, sec * 1e3, n * 4.0 * F * N * K / sec / 1e12), flush=True)
    ops.set_matrix_mode('f32')
