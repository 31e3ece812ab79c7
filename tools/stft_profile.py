"""STFT-magnitude front end timing (bench.py's stft_bench) for both reference sizes."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
import __graft_entry__ as G; G.build()
dev = torch.device('cuda:0')
for N, hop in ((1024, 256), (512, 128), (2048, 512)):
    print(json.dumps(bm.stft_bench(torch, dev, N=N, hop=hop)))
