"""In-kernel timeline of the two cell kernels (measurement build only):
    DRNMF_TIMELINE=1 python dr-nmf_amd/build.py --force && DRNMF_TIMELINE=1 python tools/timeline.py B T F r K [f16]
s_memtime stamps of wave 0 of every workgroup of the LAST launch of cell_b (kernel 0) and cell_a
(kernel 1): 0 entry, 1 first operand loads issued, 2 first chunk's MFMAs issued, 3 MFMA loop done,
4 after the cross-wave barrier, 5 stores issued.  Prints, per kernel, the span from the earliest
entry to the latest exit and the median deltas between stamps (shader clocks ~ 2.4 GHz... the
counter runs at 100 MHz on gfx950: deltas are printed in ns)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import layers, _capi
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
dev = torch.device('cuda:0')
B, T, F, r, K = [int(v) for v in sys.argv[1:6]]
f16 = 'f16' in sys.argv[6:]
N = 2 * r
W, log_h0, X = bm.synth_on_device(torch, dev, B, T, F, r, seed=1)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W, alph=0.4 * r, lam1=1.0,
         params_trainable=["log_D", "log_alph"], params_untied=["log_D", "log_alph"],
         operand_dtype='float16' if f16 else 'float32')
model = layers.build_unfolded_snmf(p, device=dev)
out = torch.empty((B, T, N), device=dev)
for _ in range(2):
    model.cell.call(X, mask_value=-1., out=out)
torch.cuda.synchronize()
L = _capi.lib()._handle if False else C.CDLL(_capi.LIB_PATH)
buf = np.zeros((2, 1024, 8), np.uint64)
rc = L.drnmf_debug_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
assert rc == 0
for kid, name in ((0, 'cell_b'), (1, 'cell_a')):
    t = buf[kid].astype(np.int64)
    live = t[:, 0] > 0
    t = t[live]
    if not len(t):
        continue
    t0 = t[:, 0].min()
    tick_ns = 1.0 / 2.4   # s_memtime ticks are shader clocks (MI355X_MICROARCH.md), ~2.4 GHz under this load
    print(name, 'workgroups stamped', len(t), 'span entry->exit of the launch %.0f ns' % ((t[:, 5].max() - t0) * tick_ns),
          'entry spread %.0f ns' % ((t[:, 0].max() - t0) * tick_ns))
    steps = ((0, 1, 'entry -> operand loads issued'), (1, 2, '-> first chunk consumed (first data)'),
             (2, 3, '-> MFMA loop done'), (3, 4, '-> reduce barrier passed'), (4, 5, '-> stores issued'),
             (0, 5, 'entry -> exit'))
    if kid == 1:
        steps = ((0, 1, 'entry -> operand loads issued'), (1, 3, '-> MFMA loop done'),
                 (3, 4, '-> reduce barrier passed'), (4, 6, '-> partial sums read from LDS'),
                 (6, 7, '-> odd-bin term'), (7, 5, '-> update, stores issued'), (0, 5, 'entry -> exit'))
    for a, b, what in steps:
        d = (t[:, b] - t[:, a]) * tick_ns
        print('   %-40s median %6.0f ns   p90 %6.0f   max %6.0f' % (what, np.median(d), np.percentile(d, 90), d.max()))
