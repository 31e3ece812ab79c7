"""Per-kernel HIP-event times of the cell at an arbitrary shape (measurement aid):
    python tools/profile_shape.py B T F r K [f16]     # e.g. 64 8 1025 4000 6 f16
Prints the launch-by-launch event times of the two cell kernels (plain launches: inflated by the
event records, good for the a/b split) and the mean launch time of the hipGraph replay of the same
forward (the figure bench.py reports)."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import layers, ops
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
dev = torch.device('cuda:0')
B, T, F, r, K = [int(v) for v in sys.argv[1:6]]
f16 = 'f16' in sys.argv[6:]
tied = 'tied' in sys.argv[6:]
N = 2 * r
W, log_h0, X = bm.synth_on_device(torch, dev, B, T, F, r, seed=1)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W,
         alph=0.4 * r, lam1=1.0, params_trainable=["log_D", "log_alph"],
         params_untied=[] if tied else ["log_D", "log_alph"],
         operand_dtype='float16' if f16 else 'float32')
model = layers.build_unfolded_snmf(p, device=dev)
cell = model.cell
cell.prepare(B, T)
desc = cell._desc(B, T)
out = torch.empty((B, T, N), device=dev)
ws = ops.cell_workspace(desc, dev)
try:
    res = ops.cell_profile(X, -1.0, cell._params_block, desc, cell.log_h0, cell._u, out, ws, frames=min(T, 6))
except ValueError as e:      # the Gram form has no launch-by-launch mode
    res = {}
for _ in range(2):
    cell.call(X, mask_value=-1., out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); cell.call(X, mask_value=-1., out=out); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (T * (2 * K - 1))      # per launch of the FACTORED form (2K-1 per frame)
fl = 2.0 * B * F * N
env = {k: v for k, v in os.environ.items() if k.startswith('DRNMF_')}
print(sys.argv[1:], env, {k: round(v, 2) for k, v in res.items()}, 'graph replay: %.2f us/launch = %.1f TF, %.0f frames/s'
      % (us, fl / us / 1e6, B * 1e6 / (us * (2 * K - 1))))
