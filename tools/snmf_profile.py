"""Dictionary training (sparse_nmf_gpu.m:210-298) at an arbitrary shape, for rocprofv3 --kernel-trace --stats:
    python tools/snmf_profile.py n F r iters [kl|ed|both] [f32|bf16x3]
One warm-up pair of iterations, then `iters` timed multiplicative-update iterations (W and H) per divergence."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import ops
n, F, r, iters = [int(v) for v in sys.argv[1:5]]
which = sys.argv[5] if len(sys.argv) > 5 else 'both'
ops.set_matrix_mode(sys.argv[6] if len(sys.argv) > 6 else 'f32')
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(6)
V = torch.rand((n, F), generator=g, device=dev) ** 2 + 1e-3
for name, beta in (('kl', 1.0), ('ed', 2.0)):
    if which not in ('both', name):
        continue
    W0 = torch.rand((F, r), generator=g, device=dev)
    H0 = torch.rand((n, r), generator=g, device=dev)
    tr = ops.SnmfTrainer(V, W0, H0, beta=beta)
    log = torch.zeros((iters + 2, 2), dtype=torch.float32, device=dev)
    for i in range(2):
        tr.step(5.0, None, True, obj=log[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        tr.step(5.0, None, True, obj=log[2 + i])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    ngemm = 4 if beta == 1.0 else 6
    print(sys.argv[1:5], name, '%.3f ms per iteration, %.1f TFLOP/s over the %d products (2 n F r each)'
          % (ms, ngemm * 2.0 * n * F * r / ms / 1e9, ngemm), 'cost %.6g -> %.6g' % (float(log[0, 1]), float(log[-1, 1])))
