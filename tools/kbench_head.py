#!/usr/bin/env python3
"""GPU time of the student CenterHead (shared conv + 6 tasks x 6 separate heads) forward+backward (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch
from distill_bev_amd import bn_act as BA

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
head = tr.detector.pts_bbox_head
x = torch.randn((8, 256, 128, 128), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def run():
    outs = head([x])
    s = sum(v.sum() for t in outs for v in t[0].values())
    s.backward()
    x.grad = None
    head.zero_grad(set_to_none=True)


def timeit(fn, n=10, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print(f"CenterHead fwd+bwd (incl. 36 output sums): {timeit(run):.2f} ms")
with torch.no_grad():
    print(f"CenterHead fwd only: {timeit(lambda: head([x])):.2f} ms")
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
ka = sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)
for e in ka[:14]:
    print(f"   {e.key[:90]:90s} {e.self_device_time_total/1e3:7.2f} ms n={e.count}")
