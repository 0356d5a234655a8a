#!/usr/bin/env python3
"""which training-mode norms of the step still take their own statistics pass (dev tool): shapes and calling lines of bn_act calls
without producer statistics (`pre is None`)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd import bn_act as BA
from distill_bev_amd.train_step import Trainer, build_model, make_batch
dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
tr.step(batch)
cnt = collections.Counter()
orig = BA._BNActTrain.forward
def fwd(ctx, x, residual, weight, bias, rm, rv, nbt, mom, eps, relu, pre=None, fork=False):
    if pre is None:
        site = "?"
        for f in reversed(traceback.extract_stack()[:-1]):
            if "/distill_bev_amd/" in f.filename and "bn_act.py" not in f.filename:
                site = f"{os.path.basename(f.filename)}:{f.lineno}"; break
        cnt[(tuple(x.shape), site)] += 1
    return orig(ctx, x, residual, weight, bias, rm, rv, nbt, mom, eps, relu, pre, fork)
BA._BNActTrain.forward = staticmethod(fwd)
tr.step(batch)
torch.cuda.synchronize()
tot = 0
for (shp, site), n in sorted(cnt.items(), key=lambda kv: -kv[1] * np.prod(kv[0][0])):
    mb = np.prod(shp) * 4 / 1e6
    tot += n * mb
    print(f"{n:3d} x {str(shp):26s} {mb:8.1f} MB  {site}")
print(f"{sum(cnt.values())} statistics passes, {tot / 1e3:.2f} GB read per step")
