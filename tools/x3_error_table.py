"""Error of the split-operand matrix mode against fp64, beside the exact-fp32 mode's, on the same inputs:
    python tools/x3_error_table.py > profiles/r06_x3_error_table.md
Frame-parallel ISTA / MU inference / dictionary training through the C ABI, the mask head, and the
time-batched weight gradients of a training step (against torch-CPU fp64 autograd of the oracle)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
from oracle import drnmf_oracle as O
dev = torch.device('cuda:0')
rows = []


def both(fn):
    out = {}
    for mode in ('f32', 'bf16x3'):
        ops.set_matrix_mode(mode)
        out[mode] = fn()
    ops.set_matrix_mode('f32')
    return out


def err(x, ref):
    d = x.double() - ref
    s = ref.abs().max().item()
    return d.abs().max().item() / s, d.pow(2).mean().sqrt().item() / s


def ista_case(n, F, N, K, alph, h0, div='ed', seed=5):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    W = torch.rand((F, N), generator=g, device=dev) ** 4
    W = W / (W * W).sum(0, keepdim=True).sqrt()
    Ht = (torch.rand((n, N), generator=g, device=dev) < 0.05) * torch.rand((n, N), generator=g, device=dev) * 5.0
    X = Ht @ W.t() + 0.01 * torch.rand((n, F), generator=g, device=dev)
    H0 = torch.full((n, N), 0.1, device=dev) if h0 == 'const 0.1' else torch.rand((n, N), generator=g, device=dev) * 0.2
    Xd, Wd, H = X.double(), W.double(), H0.double()
    for _ in range(K):
        Xh = H @ Wd.t()
        R = Xd - Xh if div == 'ed' else Xd / Xh - 1.0
        H = torch.clamp(H + (R @ Wd) / alph - 1.0 / alph, min=0.0)
    res = both(lambda: err(ops.ista_forward(X, W, H0.clone(), 1.0, alph, K, divergence=div), H))
    rows.append(('ISTA-%s %d x %d x %d, K = %d, H0 %s' % (div, n, F, N, K, h0), res))


for args in [(2048, 513, 2000, 1, 400.0, 'random'), (2048, 513, 2000, 1, 400.0, 'const 0.1'), (2048, 513, 2000, 10, 400.0, 'const 0.1'),
             (2048, 513, 2000, 25, 400.0, 'random'), (1500, 257, 200, 5, 50.0, 'random'), (777, 100, 36, 4, 20.0, 'random'),
             (600, 1025, 4000, 2, 1600.0, 'random'), (2048, 513, 2000, 10, 4000.0, 'random', 'kl')]:
    ista_case(*args)

# MU inference (sparse_nmf_gpu.m:210-229), 30 iterations
rng = np.random.default_rng(6)
n, F, N = 1500, 257, 200
W = rng.random((F, N)).astype(np.float32) * 3
V = (W @ ((rng.random((N, n)) < 0.3) * rng.random((N, n))) + 1e-3).astype(np.float32)
H0 = rng.random((N, n)).astype(np.float32)
Hr, Wr = O.mu_infer(V.astype(np.float64), W.astype(np.float64), H0.astype(np.float64), 0.1, 30, beta=2.0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
rows.append(('MU inference 1500 x 257 x 200, 30 iterations', both(lambda: err(ops.mu_forward(t(V.T), t(W), t(H0.T), 0.1, 30, beta=2.0)[0], torch.from_numpy(Hr.T).to(dev)))))

# weight gradients of a training step against fp64 autograd (the training suite's B = 250, r = 1000 case)
import test_gpu_train as T
cfg = dict(B=250, T=2, F=513, r=1000, K=2, untied=("log_D", "log_alph"))
model, P, wmask = T._setup(**cfg)
model.compile(lr=1e-3)
ref_loss, ref, cnt = T._autograd(model, P, wmask, cfg['K'], False)


def grads():
    model.loss_and_grads(t(P["X"]), t(P["Y"]), t(wmask))
    torch.cuda.synchronize()
    out = {}
    for nme, _ in model._train_items:
        g = model._gview[nme].double().cpu()
        r_ = torch.from_numpy(ref[{"kernel_clean": "kc", "kernel_noise": "kn"}.get(nme, nme)])
        out[nme] = ((g - r_).norm() / r_.norm()).item()
    return (max(out.values()), float(np.sqrt(np.mean(np.square(list(out.values()))))))
rows.append(('training step B = 250, F = 513, N = 2000, K = 2: gradient tensors, relative L2 error vs fp64 autograd (worst tensor, rms over tensors)', both(grads)))

print('# Split-operand (bf16x3) matrix mode: error against fp64, beside the exact-fp32 mode (same inputs, same box)\n')
print('Relative to max |reference|.  `tools/x3_error_table.py`; bar of VERDICT r5 item 1: within 2x the fp32 pipe\'s.\n')
print('| case | f32 max | f32 rms | bf16x3 max | bf16x3 rms | max ratio | rms ratio |')
print('|---|---|---|---|---|---|---|')
for name, r in rows:
    a, b = r['f32'], r['bf16x3']
    print('| %s | %.2e | %.2e | %.2e | %.2e | %.2f | %.2f |' % (name, a[0], a[1], b[0], b[1], b[0] / a[0], b[1] / a[1]))
print('''
The one row above 2x: an operand whose entries are all the SAME non-bf16 value (H0 = 0.1 everywhere) after ONE iteration --
every element then has the same three-plane split with the same negative mid plane, the alignment of those one-signed
correction products into the large accumulator adds up coherently instead of averaging out (tools/x3_bias.py: mean / rms
error = 1.0; with bf16-exact operands the mode is unbiased, profiles/r06_x3_steps.txt), and the residual X - H W^T
amplifies it.  The maximum error stays within 1.5x, the iteration contracts it (equal from K = 10 on), and with any
spread in the operand the mode is the MORE accurate of the two (six roundings per 16 contraction steps instead of sixteen).
Planes by truncation instead of round-to-nearest were measured and are worse (profiles/r06_x3_steps.txt).''')
