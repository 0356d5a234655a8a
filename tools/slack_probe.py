#!/usr/bin/env python3
"""Is the step GPU-bound everywhere?  (dev tool)  A 1 ms device-side sleep is queued at one point of the step (start of the forward pass,
end of the forward pass, end of the backward pass, or inside the backward pass behind the first / a middle / a late gradient); where the
GPU is the bottleneck the step gets 1 ms longer, where it waits for the host the sleep disappears in the slack."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.train_step import Trainer, build_model, make_batch, parse_losses

dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
# cycles of torch.cuda._sleep for ~1 ms (calibrated below)
def sleep_ms(cyc):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(cyc); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)
cyc = 1000000
cyc = int(cyc / sleep_ms(cyc))
print("sleep calibration: %.3f ms" % sleep_ms(cyc))
params = [p for p in tr.module.parameters() if p.requires_grad]
names = {id(p): n for n, p in tr.module.named_parameters()}
hook_at = {"bwd_first": params[-1], "bwd_mid": params[len(params) // 2], "bwd_late": params[5]}
where = "none"
handles = []
for tag, p in hook_at.items():
    def mk(tag):
        def h(_p):
            if where == tag:
                torch.cuda._sleep(cyc)
        return h
    handles.append(p.register_post_accumulate_grad_hook(mk(tag)))
def step():
    if where == "fwd_start": torch.cuda._sleep(cyc)
    losses = tr.module(**batch); loss = parse_losses(losses)
    if where == "fwd_end": torch.cuda._sleep(cyc)
    tr.optimizer.zero_grad(set_to_none=True); loss.backward()
    if where == "bwd_end": torch.cuda._sleep(cyc)
    torch.nn.utils.clip_grad_norm_(tr.params, **tr.grad_clip); tr.optimizer.step()
for _ in range(5): step()
res = {}
for rep in range(3):
    for w in ["none", "fwd_start", "fwd_end", "bwd_first", "bwd_mid", "bwd_late", "bwd_end"]:
        where = w
        for _ in range(2): step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize()
        res.setdefault(w, []).append((time.perf_counter() - t) / 8 * 1e3)
for w, v in res.items():
    print("%-10s %s  (vs none %+.2f ms)" % (w, " ".join("%.2f" % x for x in v), np.mean(v) - np.mean(res["none"])))
print("hook parameters:", {t: names[id(p)] for t, p in hook_at.items()})
