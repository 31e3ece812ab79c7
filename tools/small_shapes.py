#!/usr/bin/env python3
"""Small-shape lines of bench.py alone (the shapes the persistent Gram chains serve): the shipped
r = 100 training configurations and BASELINE configs[0].  usage: small_shapes.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
out = {"DRNMF_PERSIST": os.environ.get("DRNMF_PERSIST", "(default)")}
for name, shape in (("r100_K5", (32, 500, 257, 100, 5)), ("r100_K2", (32, 500, 257, 100, 2)),
                    ("r1000_K5", (32, 500, 257, 1000, 5))):
    r = bench.train_bench(torch, dev, steps=steps, warmup=3, shape=shape)
    out[name] = {k: r[k] for k in ("ms_per_step", "cell_forward_ms", "bptt_sequential_ms",
                                   "bptt_time_batched_ms", "chain_launch_us", "form")}
c1 = bench.config1_bench(torch, dev)
out["config1"] = {k: c1[k] for k in ("gpu_frames_per_s", "gpu_ms_per_utterance", "gpu_launch_us",
                                     "cpu_best_frames_per_s")}
print(json.dumps(out, indent=1))
