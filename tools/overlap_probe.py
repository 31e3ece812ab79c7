"""Measurement: does a frame-parallel GEMM stream run beside the recurrent chain?  The chain's
launches are latency-bound (20 % of the MFMA peak); if big GEMMs on a second stream fill the idle
matrix cores without slowing the chain much, the BPTT's time-batched weight gradients could hide
behind its sequential pass.  Runs the C2 forward (chain) alone, K iterations of frame-parallel ISTA
(two 128x128-tile GEMMs per iteration) alone, then both concurrently on two streams."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G; G.build()
from drnmf_amd import layers, ops
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
dev = torch.device('cuda:0')
B, T, F, r, K = 64, 400, 513, 1000, 25
N = 2 * r
W, log_h0, X = bm.synth_on_device(torch, dev, B, T, F, r, seed=1)
p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K, W=W, alph=400.0,
         lam1=1.0, params_trainable=["log_D", "log_alph"], params_untied=["log_D", "log_alph"])
model = layers.build_unfolded_snmf(p, device=dev)
out = torch.empty((B, T, N), device=dev)
n = 32768
g = torch.Generator(device=dev); g.manual_seed(1)
Wt = torch.from_numpy(W).to(dev)
Xf = torch.rand((n, F), generator=g, device=dev)
H = torch.full((n, N), 0.1, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def chain():
    with torch.cuda.stream(s1):
        model.cell.call(X, mask_value=-1., out=out)
def gemms(it):
    with torch.cuda.stream(s2):
        ops.ista_forward(Xf, Wt, H, 1.0, 400.0, it)
chain(); gemms(2); torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
t_chain = timed(chain)
IT = 60
t_gemm = timed(lambda: gemms(IT))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
torch.cuda.synchronize()
with torch.cuda.stream(s1):
    ev[0].record(s1)
with torch.cuda.stream(s2):
    ev[2].record(s2)
chain(); gemms(IT)
with torch.cuda.stream(s1):
    ev[1].record(s1)
with torch.cuda.stream(s2):
    ev[3].record(s2)
torch.cuda.synchronize()
print('chain alone %.1f ms, %d ISTA iterations alone %.1f ms; together: chain %.1f ms, GEMMs %.1f ms'
      % (t_chain, IT, t_gemm, ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])))
