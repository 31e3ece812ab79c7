import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import importlib.util
sp = importlib.util.spec_from_file_location('b', '/root/repo/bench.py'); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
dev = torch.device('cuda:0')
r = bm.train_bench(torch, dev, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
print(r)
