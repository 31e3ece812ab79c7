"""One training step (forward + BPTT + Adam) at an arbitrary shape, for rocprofv3 --kernel-trace --stats:
    python tools/train_profile.py B T F r K [steps]
"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
B, T, F, r, K = [int(v) for v in sys.argv[1:6]]
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
dev = torch.device('cuda:0')
import __graft_entry__ as G; G.build()
print(bm.train_bench(torch, dev, steps=steps, warmup=1, shape=(B, T, F, r, K), ragged=len(sys.argv) <= 7))
