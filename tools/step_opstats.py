#!/usr/bin/env python3
"""torch.profiler over one bench-configuration training step: GPU time of selected ATen ops grouped by input
shape (dev tool: finds layout copies, unfused norm layers, torch upsample calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
names = ("aten::copy_", "aten::clone", "aten::contiguous", "aten::miopen_batch_norm", "aten::miopen_batch_norm_backward",
         "aten::add", "aten::add_", "aten::upsample_bilinear2d", "aten::upsample_bilinear2d_backward", "aten::native_batch_norm",
         "aten::mul", "aten::sum", "aten::cat", "aten::relu", "aten::relu_", "aten::threshold_backward", "aten::sigmoid")
allk = sorted(ka, key=lambda e: -e.self_device_time_total)
agg = {}
for e in allk:
    agg.setdefault(e.key, [0, 0])
    agg[e.key][0] += e.self_device_time_total; agg[e.key][1] += e.count
print("top ops by self device time (us/step, calls):")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"   {k[:70]:70s} {t:9.0f} us  n={n}")
rows = [e for e in ka if e.key in names]
rows.sort(key=lambda e: -e.device_time_total)
tot = {}
for e in rows:
    tot[e.key] = tot.get(e.key, 0) + e.device_time_total
print("totals (us per step):", {k: round(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
for e in rows[:60]:
    print(f"{e.key:38s} n={e.count:4d} gpu {e.device_time_total:9.0f} us  shapes {str(e.input_shapes)[:110]}")
print("host time of the step (ms):", sum(e.self_cpu_time_total for e in ka) / 1e3)
