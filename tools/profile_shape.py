"""Per-kernel HIP-event times of the cell at an arbitrary shape (measurement aid):
    python tools/profile_shape.py B T F r K        # e.g. 64 8 1025 4000 6
"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import layers, ops
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
dev = torch.device('cuda:0')
B, T, F, r, K = [int(v) for v in sys.argv[1:6]]
N = 2 * r
W, log_h0, X = bm.synth_on_device(torch, dev, B, T, F, r, seed=1)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
         alph=0.4 * r, lam1=1.0, params_trainable=["log_D", "log_alph"], params_untied=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
cell = model.cell
cell.prepare(B, T)
desc = cell._desc(B, T)
out = torch.empty((B, T, N), device=dev)
ws = ops.cell_workspace(desc, dev)
res = ops.cell_profile(X, -1.0, cell._params_block, desc, cell.log_h0, cell._u, out, ws, frames=min(T, 6))
fl = 2.0 * B * F * N
print(sys.argv[1:], os.environ.get('DRNMF_RB'), res, 'TF a/b: %.1f %.1f' % (fl / res['cell_a_us'] / 1e6, fl / res['cell_b_us'] / 1e6))
