#!/usr/bin/env python3
"""Where do the layout copies of the step come from?  torch.profiler with Python stacks over one step: every aten::contiguous /
aten::clone that moved more than 1 MB, with the innermost frames of this repository (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench_workloads as BW

dev = torch.device("cuda:0")
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
wl = BW.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "distill_step"](dev, 0, 1)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    wl.step()
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    if e.name not in ("aten::contiguous", "aten::clone") or not e.input_shapes or not e.input_shapes[0]:
        continue
    n = 1
    for d in e.input_shapes[0]:
        n *= d
    if n * 4 < (1 << 20) or e.device_time_total <= 0:
        continue
    frames = [f for f in (e.stack or []) if "site-packages" not in f and "dist-packages" not in f and "<built-in" not in f][:4]
    key = (e.name, tuple(e.input_shapes[0]), tuple(frames))
    seen.setdefault(key, [0, 0.0]); seen[key][0] += 1; seen[key][1] += e.device_time_total
for (name, shp, frames), (cnt, us) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{us:8.0f} us {cnt:3d}x {name} {list(shp)}")
    for f in frames:
        print("            ", f[:150])
