"""Gradient error against fp64 autograd as the sequence grows (diagnostic behind
tests/test_gpu_fullsize.py::test_config3_r1000_*):  python tools/grad_vs_T.py r K T [T ...]
For each T: relative L2 error of every gradient tensor of (a) the GPU training step and (b) torch-CPU float32
autograd of the oracle restatement -- the same arithmetic precision on the host -- both against torch-CPU float64."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G; G.build()
import test_gpu_train as TT
dev = torch.device("cuda:0")
r, K = int(sys.argv[1]), int(sys.argv[2])
for T in [int(v) for v in sys.argv[3:]]:
    cfg = dict(B=32, T=T, F=257, r=r, K=K, untied=("log_D", "log_alph"))
    model, P, wmask = TT._setup(**cfg)
    model.compile(lr=1e-3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    flat = model.loss_and_grads(t(P["X"]), t(P["Y"]), t(wmask)).clone()
    torch.cuda.synchronize()
    l64, g64, _ = TT._autograd(model, P, wmask, K, False)
    l32, g32, _ = TT._autograd(model, P, wmask, K, False, dtype=torch.float32)
    name_map = {"kernel_clean": "kc", "kernel_noise": "kn"}
    print("T=%d  loss rel err: gpu %.2e  host-f32 %.2e" % (T, abs(float(flat[-4]) - l64) / abs(l64), abs(l32 - l64) / abs(l64)))
    for n, _ in model._train_items:
        g = model._gview[n].cpu().numpy().astype(np.float64)
        ref = g64[name_map.get(n, n)]; h32 = g32[name_map.get(n, n)]
        nr = max(np.linalg.norm(ref), 1e-30); sc = max(np.max(np.abs(ref)), 1e-30)
        print("   %-14s relL2 gpu %.2e host-f32 %.2e | max gpu %.2e host-f32 %.2e | gpu-vs-host-f32 relL2 %.2e" % (
            n, np.linalg.norm(g - ref) / nr, np.linalg.norm(h32 - ref) / nr, np.max(np.abs(g - ref)) / sc,
            np.max(np.abs(h32 - ref)) / sc, np.linalg.norm(g - h32) / nr))
    del model
    torch.cuda.empty_cache()
