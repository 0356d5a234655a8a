#!/usr/bin/env python3
"""N back-to-back bench-configuration steps and nothing else (profiling target: tools/gpu_gaps.py, rocprofv3).  usage: steps_only.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.train_step import Trainer, build_model, make_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n):
    tr.step(batch)
torch.cuda.synchronize()
print(f"{n} back-to-back steps: {(time.perf_counter() - t) / n * 1e3:.2f} ms/step")
