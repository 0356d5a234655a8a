#!/usr/bin/env python3
"""Host-side issue time vs GPU time of the bench-configuration step (dev tool): is the step launch-bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd.train_step import Trainer, build_model, make_batch

dev = torch.device("cuda:0")
if os.environ.get("DBEV_FORCE_DDP") == "1":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
for i in range(6):
    t = time.perf_counter()
    tr.step(batch)
    host = time.perf_counter() - t
    torch.cuda.synchronize()
    tot = time.perf_counter() - t
    print(f"step {i}: host issue {host*1e3:7.1f} ms   wall {tot*1e3:7.1f} ms")
# back-to-back steps (what the bench measures)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(8):
    tr.step(batch)
torch.cuda.synchronize()
print(f"8 back-to-back steps: {(time.perf_counter()-t)/8*1e3:7.1f} ms/step")
# phase split of the host time: forward issue / backward issue / optimizer issue
for _ in range(2):
    t0 = time.perf_counter(); losses = tr.module(**batch); from distill_bev_amd.train_step import parse_losses
    loss = parse_losses(losses); t1 = time.perf_counter()
    tr.optimizer.zero_grad(set_to_none=True); loss.backward(); t2 = time.perf_counter()
    if tr.grad_clip:
        torch.nn.utils.clip_grad_norm_(tr.params, **tr.grad_clip)
    tr.optimizer.step(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"host: forward {1e3*(t1-t0):6.1f}  backward {1e3*(t2-t1):6.1f}  clip+opt {1e3*(t3-t2):6.1f}  drain {1e3*(t4-t3):6.1f} ms")
sys.exit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    tr.step(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
