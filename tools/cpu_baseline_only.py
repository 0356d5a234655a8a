#!/usr/bin/env python3
"""Time only the cpu_baseline leg of the distill_step workload (dev tool; no GPU work)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_workloads as W
wl = W.DistillStep.__new__(W.DistillStep)
wl.N_POINTS = W.DistillStep.N_POINTS
t = time.time()
print(wl.cpu_baseline(), "wall %.1f s" % (time.time() - t))
