#!/usr/bin/env python3
"""Which lines of this repository make device copies in one training step (dev tool): Tensor.contiguous / clone / to / float / cat /
stack / zeros / full ... are wrapped and every call that produced a new device buffer is logged with its innermost repository frame."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
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
log = collections.defaultdict(lambda: [0, 0])
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "/distill_bev_amd/" in f.filename:
            return f"{f.filename.split('/distill_bev_amd/')[-1]}:{f.lineno}"
    return "?"
def wrap_method(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        r = orig(self, *a, **k)
        if isinstance(r, torch.Tensor) and r.is_cuda and (not self.is_cuda or r.data_ptr() != self.data_ptr()) and r.numel() > 0:
            e = log[(name, site())]; e[0] += 1; e[1] += r.numel() * r.element_size()
        return r
    setattr(torch.Tensor, name, f)
def wrap_fn(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        r = orig(*a, **k)
        if isinstance(r, torch.Tensor) and r.is_cuda and r.numel() > 0:
            e = log[(name, site())]; e[0] += 1; e[1] += r.numel() * r.element_size()
        return r
    setattr(mod, name, f)
for n in ("contiguous", "clone", "to", "float", "permute"):
    if n != "permute":
        wrap_method(n)
for n in ("cat", "stack", "zeros", "zeros_like", "full", "ones", "ones_like", "arange", "linspace", "tensor"):
    wrap_fn(torch, n)
tr.step(batch)
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: -kv[1][1])
print(f"# {sum(v[0] for v in log.values())} logged calls, {sum(v[1] for v in log.values()) / 1e6:.1f} MB")
for (name, s), (n, b) in rows[:60]:
    print(f"{n:4d} {b / 1e6:9.2f} MB  {name:12s} {s}")
