#!/usr/bin/env python3
"""Run a few distillation training steps on one GPU and print losses / timing (dev tool)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from distill_bev_amd.train_step import Trainer, build_model, make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--points", type=int, default=240000)
ap.add_argument("--channels-last", action="store_true")
ap.add_argument("--benchmark", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = args.benchmark
t0 = time.time()
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=args.channels_last)
print("build+to(device) %.1fs" % (time.time() - t0), flush=True)
batch = make_batch(args.bs, np.random.default_rng(0), dev, n_points=args.points)

torch.cuda.synchronize()
for i in range(args.steps):
    t = time.time()
    loss, losses = tr.step(batch)
    host = time.time() - t
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"step {i}: loss {float(loss):.4f}  {dt*1e3:.1f} ms (host issue {host*1e3:.1f} ms)  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    if i == 0:
        for k, v in losses.items():
            print(f"   {k}: {float(v):.5f}")
        assert all(torch.isfinite(v) for v in losses.values())
