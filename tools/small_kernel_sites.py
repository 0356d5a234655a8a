#!/usr/bin/env python3
"""Which Python lines launch the step's small fill / copy / memset kernels (dev tool): torch.profiler with stacks over one step,
aten::fill_ / aten::zero_ / aten::copy_ / aten::zeros ... grouped by the innermost frame of this repository."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::full", "aten::ones", "aten::clone",
        "aten::_to_copy", "aten::contiguous", "aten::arange", "aten::add", "aten::add_")
sites = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=24):
    if ev.key not in want or ev.device_time_total <= 0:
        continue
    own = [f for f in (ev.stack or []) if "/distill_bev_amd/" in f or "bench" in f]
    key = (ev.key, own[0].split("/distill_bev_amd/")[-1] if own else ((ev.stack or ["?"])[0][-70:]))
    sites[key][0] += ev.count
    sites[key][1] += ev.self_device_time_total
tot = collections.defaultdict(lambda: [0, 0.0])
for (name, site), (n, t) in sites.items():
    tot[name][0] += n; tot[name][1] += t
for name, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"== {name}: {n} device launches, {t / 1e3:.3f} ms")
for (name, site), (n, t) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{n:4d} {t / 1e3:8.3f} ms  {name:16s} {site}")
