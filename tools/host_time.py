#!/usr/bin/env python3
"""Host-side issue time vs GPU time of the bench-configuration step (dev tool): is the step launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model()
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
for i in range(6):
    t = time.perf_counter()
    tr.step(batch)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    tot = time.perf_counter() - t
    print(f"step {i}: host issue {host*1e3:7.1f} ms   wall {tot*1e3:7.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    tr.step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
