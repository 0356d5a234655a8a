#!/usr/bin/env python3
"""Is the forward half of the step host-bound?  Per step: host time to ISSUE the forward / backward+optimizer, and the GPU time between
the events recorded around each (dev tool).  GPU span ~= host span -> the host is the bottleneck of that phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch import nn
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.train_step import Trainer, build_model, make_batch, parse_losses

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
rows = []
for it in range(6):
    e = [ev() for _ in range(4)]
    t0 = time.perf_counter(); e[0].record()
    losses = tr.module(**batch)
    loss = parse_losses(losses)
    e[1].record(); t1 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    e[2].record(); t2 = time.perf_counter()
    if tr.grad_clip:
        nn.utils.clip_grad_norm_(tr.params, **tr.grad_clip)
    tr.optimizer.step()
    if tr.packer is not None:
        tr.packer.repack()
    e[3].record(); t3 = time.perf_counter()
    rows.append((e, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
torch.cuda.synchronize()
for e, hf, hb, ho in rows:
    print(f"host issue: fwd {hf:6.1f}  bwd {hb:6.1f}  opt {ho:6.1f} ms | GPU: fwd {e[0].elapsed_time(e[1]):6.1f}  bwd {e[1].elapsed_time(e[2]):6.1f}  opt {e[2].elapsed_time(e[3]):6.1f} ms")
g = getattr(tr.detector, "adjacent_graph", None)
if g is not None:
    print("adjacent graph: captures", g.captures, "replays", g.replays, "eager", g.eager)
