#!/usr/bin/env python3
"""Inference forward (cell + mask head) of the headline model over batch sizes: frames/s and the
fraction of the fp32-MFMA peak (bench.slab_bench on B utterances x T frames).
usage: batch_sweep.py [T] [B ...]      (DRNMF_SPLIT / DRNMF_RB / DRNMF_KS select variants;
       DRNMF_SWEEP_SHAPE="F r K" another model, e.g. "257 1000 5" = the shipped r = 1000 configuration)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Bs = [int(v) for v in sys.argv[2:]] or [64, 128, 250, 512, 1024]
env = {k: v for k, v in os.environ.items() if k.startswith("DRNMF_")}
out = {"T": T, "env": env, "rows": []}
F_, r_, K_ = (int(v) for v in os.environ.get("DRNMF_SWEEP_SHAPE", "513 1000 25").split())
for B in Bs:
    r = bench.slab_bench(torch, dev, F_, r_, K_, T, slab=B)
    out["rows"].append({"B": B, "frames_per_s": round(r["frames_per_s"]), "frac": round(r["frac_of_f32_mfma_peak"], 4),
                        "ms": round(r["ms_per_slab"], 2)})
    print(json.dumps(out["rows"][-1]), env, flush=True)
