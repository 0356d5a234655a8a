#!/usr/bin/env python3
"""torch.profiler over one bench-configuration training step: GPU time of every convolution forward / backward grouped by
input shape, with FLOPs and the max(MFMA, HBM) bound of each shape (dev tool: which layers of the step are far from their bound).

    python tools/step_convstats.py [workload]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench_workloads as BW

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "distill_step"
wl = BW.WORKLOADS[name](dev, 0, 1)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    wl.step()
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
print(f"{name}: convolution ops {tot / 1e3:.2f} ms of device time in one step")
gemm = {}
for e in ka:
    if e.key in ("aten::addmm", "aten::mm", "aten::bmm", "aten::linear", "aten::matmul") and e.self_device_time_total > 0:
        gemm.setdefault(e.key, [0, 0]); gemm[e.key][0] += e.self_device_time_total; gemm[e.key][1] += e.count
print("GEMM ops (self device ms, calls):", {k: (round(v[0] / 1e3, 2), v[1]) for k, v in gemm.items()})
for t, c, k, s, _, _ in sorted(rows, key=lambda r: -r[0])[:70]:
    print(f"{t / 1e3:8.3f} ms {c:4d}x {k.replace('aten::', ''):24s} {s}")
