"""dev tool: per-launch breakdown of one logged kernel family inside the distillation step: launches grouped by their work field (FLOPs or
bytes -> one group per layer shape), with count per step, mean us and rate.   python tools/klog_breakdown.py b6_fwd [b6_wgrad ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
import bench_workloads as W
from distill_bev_amd import _lib as L
names = sys.argv[1:] or ["b6_fwd"]
dev = torch.device("cuda:0")
wl = W.DistillStep(dev, 0, 1)
for _ in range(4):
    wl.step()
torch.cuda.synchronize()
L.kernel_timing_read(); L.kernel_timing(names)
N = 3
for _ in range(N):
    wl.step()
log = L.kernel_timing_read(); L.kernel_timing(False)
for k in names:
    recs = log.get(k, [])
    groups = collections.OrderedDict()
    for ms, work in recs:
        groups.setdefault(work, []).append(ms)
    tot = sum(ms for ms, _ in recs) / N
    print(f"{k}: {len(recs) / N:.0f} launches, {tot:.3f} ms per step")
    for work, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        t = sum(v) / len(v)
        print(f"   work {work / 1e9:9.2f} G  x{len(v) / N:5.1f}/step  avg {t * 1e3:8.1f} us  -> {work / t / 1e9:8.1f} G/s-units(T)  share {sum(v) / N:6.3f} ms")
