#!/usr/bin/env python3
"""torch.profiler over one bench-configuration training step: GPU time of every convolution forward / backward grouped by
input shape, with FLOPs and the max(MFMA, HBM) bound of each shape (dev tool: which layers of the step are far from their bound).

    python tools/step_convstats.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model()
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = []
for e in ka:
    if e.key not in ("aten::miopen_convolution", "aten::convolution_backward", "aten::miopen_convolution_transpose", "aten::cudnn_convolution",
                     "aten::_convolution"):
        continue
    if e.key == "aten::_convolution" and e.self_device_time_total == 0:
        continue
    shp = e.input_shapes
    try:
        if e.key == "aten::convolution_backward":
            gy, x, w = shp[0], shp[1], shp[2]
        else:
            x, w = shp[0], shp[1]
            gy = None
        N, ci, H, W = x
        co, cig, kh, kw = w
        if gy is not None:
            ho, wo = gy[2], gy[3]
        else:
            ho = wo = None
    except Exception:
        rows.append((e.self_device_time_total, e.count, e.key, str(shp), 0, 0)); continue
    rows.append((e.self_device_time_total, e.count, e.key, f"x{x} w{w}" + (f" gy{gy}" if gy else ""), 0, 0))
tot = sum(r[0] for r in rows)
print(f"convolution ops: {tot / 1e3:.2f} ms of device time in one step")
for t, c, k, s, _, _ in sorted(rows, key=lambda r: -r[0])[:70]:
    print(f"{t / 1e3:8.3f} ms {c:4d}x {k.replace('aten::', ''):24s} {s}")
