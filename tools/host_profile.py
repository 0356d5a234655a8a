#!/usr/bin/env python3
"""cProfile of the host side of the bench-configuration step (dev tool): where do the ~138 ms of issue time go?"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
N = 6
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"# {N} steps; times below are totals over them")
st.print_stats(45)
st.sort_stats("cumulative")
st.print_stats(60)
