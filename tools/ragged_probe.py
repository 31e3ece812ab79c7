"""Where does length-aware predict spend its time?  (tools, not product)"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
from drnmf_amd import layers
dev = torch.device('cuda:0')
F, r, K, T, n, slab = [int(v) for v in (sys.argv[1:7] if len(sys.argv) > 6 else (257, 1000, 5, 2000, 2000, 250))]
N = 2 * r
W, log_h0, X = bm.synth_on_device(torch, dev, slab, T, F, r, seed=123)
rng = np.random.Generator(np.random.PCG64(7654))
lens = rng.integers(int(0.4 * T), T + 1, size=n)
xh = np.concatenate([X.cpu().numpy()] * ((n + slab - 1) // slab))[:n].copy()
for i, L in enumerate(lens):
    xh[i, L:] = -1.0
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W, alph=400.0 if r >= 1000 else 50.0,
         lam1=1.0, params_trainable=["log_D", "log_alph"], params_untied=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
valid = int(lens.sum())
t0 = time.perf_counter(); l2 = model.valid_lengths(xh, -1.0); print('valid_lengths %.3f s' % (time.perf_counter() - t0), np.array_equal(l2, lens))
ref = None
for name, kw in (("padded", dict(length_aware=False)), ("aware", dict()), ("aware+lengths", dict(lengths=lens))):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = model.predict(xh, batch_size=slab, **kw)
        sec = time.perf_counter() - t0
        print('%-14s rep %d: %.3f s = %.0f valid frames/s' % (name, rep, sec, valid / sec), flush=True)
    if ref is None:
        ref = m
    else:
        d = max(float(np.abs(m[i, :L] - ref[i, :L]).max()) for i, L in enumerate(lens) if L)
        print('   max |diff| on valid frames vs padded run: %.3g' % d)
# device-only forward at the two slab shapes
order = np.argsort(-lens, kind='stable')
for s in range(0, n, slab):
    idx = order[s:s + slab]; Ts = int(-(-int(lens[idx].max()) // 32) * 32)
    xd = torch.from_numpy(np.ascontiguousarray(xh[idx][:, :Ts])).to(dev)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); model.forward(xd); torch.cuda.synchronize()
        print('device forward slab %d (%d x %d): %.3f s' % (s // slab, len(idx), Ts, time.perf_counter() - t0))
xd = torch.from_numpy(xh[:slab]).to(dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); model.forward(xd); torch.cuda.synchronize()
    print('device forward padded slab (%d x %d): %.3f s' % (slab, T, time.perf_counter() - t0))
