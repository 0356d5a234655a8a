#!/usr/bin/env python3
"""Host/GPU synchronisation points of the bench-configuration step (dev tool): torch.cuda.set_sync_debug_mode("warn") over one
step after warm-up; prints each distinct call site once."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
import traceback
seen = {}
def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "/distill_bev_amd/" in f.filename or "bench" in f.filename]
    key = tuple((f.filename.split("/")[-1], f.lineno) for f in st[-3:])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
tr.step(batch)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, " <- ".join(f"{a}:{b}" for a, b in reversed(k)))
