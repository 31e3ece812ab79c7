"""A/B of the thin last column tile of gemm_nt (csrc/gemm_nt.h THIN_COLS; DRNMF_THIN=0 restores the full
pipeline / the separate tail kernels): frame-parallel ISTA-ED, MU inference, mask head at F = 513."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
import bench
from drnmf_amd import ops
dev = torch.device('cuda:0')
F, r, K = 513, 1000, 25
N = 2 * r
rng = np.random.RandomState(0)
W = (rng.rand(F, N).astype(np.float32) + 0.05)
W /= np.linalg.norm(W, axis=0, keepdims=True)
out = {"DRNMF_THIN": os.environ.get("DRNMF_THIN", "(unset)")}
out["ista"] = {k: round(v, 4) for k, v in bench.ista_bench(torch, dev, F, N, K, W).items()}
out["mu"] = bench.mu_bench(torch, dev, F, N, W)
# mask head over 128 000 rows (the headline's B*T)
g = torch.Generator(device=dev); g.manual_seed(3)
rows = 128000
h = torch.rand((rows, N), generator=g, device=dev)
kc = torch.rand((r, F), generator=g, device=dev) - 3.0
kn = torch.rand((r, F), generator=g, device=dev) - 3.0
m = torch.empty((rows, F), device=dev)
ops.head_forward(h.view(64, 2000, N), kc, kn, out=m.view(64, 2000, F)); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.head_forward(h.view(64, 2000, N), kc, kn, out=m.view(64, 2000, F))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
out["head"] = {"ms": round(ms, 3), "tflops": round(rows * 4.0 * F * N / ms / 1e9, 1)}
print(json.dumps(out))
