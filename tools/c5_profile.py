"""The config-5 shape (F=1025, N=8000, K=50 untied, B=64) forward with fp32 and fp16 operands, for
rocprofv3 --kernel-trace --stats:
    python tools/c5_profile.py [frames]
"""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
sp = importlib.util.spec_from_file_location('b', os.path.join(ROOT, 'bench.py')); bm = importlib.util.module_from_spec(sp); sp.loader.exec_module(bm)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
import __graft_entry__ as G; G.build()
print(json.dumps(bm.config5_bench(torch, torch.device('cuda:0'), frames=frames)))
