#!/usr/bin/env python3
"""Run the full-size forward twice with forward hooks on every leaf module and report the first modules whose outputs
differ bit-wise between the two runs (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))
model, cfg = build_model(seed=0, allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, world_size=1, channels_last=True)
batch = make_batch(B, np.random.default_rng(1234), dev, n_points=240000)
det = tr.detector
runs = []
def run():
    rec = []
    hs = []
    def mk(name):
        def hook(m, i, o):
            if torch.is_tensor(o):
                rec.append((name, type(m).__name__, o.detach().float().sum().item(), o.detach().clone() if o.numel() < 5e7 else o.detach().double().abs().sum().item()))
        return hook
    for root, pre in ((det, ""), (det.teacher_model, "teacher.")):
        for n, m in root.named_modules():
            if len(list(m.children())) == 0:
                hs.append(m.register_forward_hook(mk(pre + n)))
    with torch.no_grad():
        losses = det.forward_train(**batch)
    for h in hs: h.remove()
    return rec, losses
r1, l1 = run(); r2, l2 = run()
print("modules recorded", len(r1), len(r2))
n = 0
for (a, b) in zip(r1, r2):
    same = torch.equal(a[3], b[3]) if torch.is_tensor(a[3]) else a[3] == b[3]
    if not same:
        print("DIFF", a[0], a[1], a[2], b[2])
        n += 1
        if n > 12: break
print("loss diffs:", [k for k in l1 if not torch.equal(l1[k], l2[k])][:50])
