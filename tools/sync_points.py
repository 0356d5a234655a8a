#!/usr/bin/env python3
"""Host/GPU synchronisation points of one step of a bench workload (dev tool): torch.cuda.set_sync_debug_mode("warn") over one
step after warm-up; prints each distinct call site (innermost frames of this repository) with its count.
usage: sync_points.py [workload]   (default: the headline distillation step)"""
import os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
import bench_workloads as W

name = sys.argv[1] if len(sys.argv) > 1 else W.WORKLOADS["default"]
dev = torch.device("cuda:0")
wl = W.WORKLOADS[name](dev, 0, 1)
for _ in range(3):
    wl.step()
torch.cuda.synchronize()
seen = {}
def show(message, category, filename, lineno, file=None, line=None):
    st = traceback.extract_stack()[:-1]
    own = [f for f in st if "/distill_bev_amd/" in f.filename]
    fr = own[-3:] if own else st[-6:]
    key = tuple((f.filename.split("/")[-1], f.lineno) for f in fr)
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
wl.step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"# {name}: {sum(seen.values())} synchronising calls in one step")
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, " <- ".join(f"{a}:{b}" for a, b in reversed(k)))
