#!/usr/bin/env python3
"""per-op timings of the hot-path operators -- the reference's CPU path (the compiled reference voxelizer where it
exists, else the oracle restatement of the reference's op sequence) on the host cores, beside the HIP kernels on the GPU, at the
step's per-sample sizes.  Test infrastructure: imports oracle/.   python tools/cpu_op_timings.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from distill_bev_amd import synthetic as syn
from distill_bev_amd import voxel as V
from distill_bev_amd.distill_loss import ForegroundMaskRasterizer
from oracle import distill as OD
from oracle import lss_torch as OT
from oracle import voxel as OV

dev = torch.device("cuda:0")
ncores = min(64, os.cpu_count() or 1)
torch.set_num_threads(ncores)
rng = np.random.default_rng(0)
print(f"host: {os.cpu_count()} hardware threads, {ncores} used by torch; GPU: {torch.cuda.get_device_name(0)}")


def cpu_time(fn, budget=5.0, max_reps=50):
    fn()
    reps, t0 = 0, time.perf_counter()
    while True:
        fn(); reps += 1
        if time.perf_counter() - t0 > budget or reps >= max_reps:
            break
    return (time.perf_counter() - t0) / reps * 1e3


def gpu_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


rows = []
# ---- dynamic voxelization, one 240 k-point sweep (reference: ops/voxel/src/voxelization_cpu.cpp, compiled into oracle/_ref) --------
pts = syn.lidar_points(240000, rng)
vs, pcr = [0.2, 0.2, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
from oracle.build_ref import load_ref
ref = load_ref()
pt = torch.from_numpy(pts)
if ref is not None:
    co = torch.zeros((pts.shape[0], 3), dtype=torch.int32)
    t_cpu = cpu_time(lambda: ref.dynamic_voxelize(pt, co, vs, pcr, 3)); kind = "reference C++ (1 thread)"
else:
    t_cpu = cpu_time(lambda: OV.dynamic_voxelize(pts, vs, pcr)); kind = "oracle/voxel.c (1 thread)"
pd = pt.to(dev); cd = torch.zeros((pts.shape[0], 3), dtype=torch.int32, device=dev)
rows.append(("dynamic_voxelize, 240 k points", kind, t_cpu, gpu_time(lambda: V.dynamic_voxelize(pd, cd, vs, pcr, 3))))

# ---- dynamic scatter (max), 240 k points x 64 features -> pillars (reference has no CPU path: oracle/voxel.c) ----------------------
coors = OV.dynamic_voxelize(pts, vs, pcr)
keep = (coors >= 0).all(1)
feats = rng.standard_normal((int(keep.sum()), 64)).astype(np.float32)
ck = np.ascontiguousarray(coors[keep])
t_cpu = cpu_time(lambda: OV.dynamic_scatter_forward(feats, ck, "max"))
fd = torch.from_numpy(feats).to(dev); cdk = torch.from_numpy(ck).to(dev)
rows.append(("dynamic_scatter max, C = 64", "oracle/voxel.c (1 thread)", t_cpu, gpu_time(lambda: V.dynamic_point_to_voxel_forward(fd, cdk, "max"))))

# ---- lift-splat (voxel_pooling), one sample = 2 six-camera frames, forward + backward ----------------------------------------------
from bench_workloads import BevPoolCfg1, _grid
wl = BevPoolCfg1(dev, 0, 1)
dx, bx, nx = _grid()
geom = wl._geom_cpu[:2]
x = torch.randn((2, 6, 59, 16, 44, 64)).requires_grad_(True)
g = torch.randn((2, 64, 128, 128))


def splat_cpu():
    x.grad = None
    OT.voxel_pooling_cumsum(geom, x, dx, bx, nx).backward(g)


t_cpu = cpu_time(splat_cpu, budget=8.0, max_reps=10)
t_gpu = gpu_time(wl.step) / (wl.B * wl.F) * 2
rows.append(("voxel_pooling fwd + bwd, 1 sample (2 frames)", f"oracle/lss_torch (argsort + cumsum trick), {ncores} threads", t_cpu, t_gpu))

# ---- foreground / scale masks of one sample (30 boxes, 128 x 128 map) ---------------------------------------------------------------
boxes, _ = syn.gt_boxes(30, rng)
t_cpu = cpu_time(lambda: OD.foreground_scale_mask(128, 128, [boxes]))
rast = ForegroundMaskRasterizer([1024, 1024, 40], pcr, [0.1, 0.1, 0.2])
lb = [boxes]
rows.append(("foreground_scale_mask, 30 boxes, 128 x 128", "oracle/distill.py (numpy, as box_np_ops)", t_cpu, gpu_time(lambda: rast(128, 128, lb, dev))))

print(f"{'op':48s} {'CPU path':62s} {'CPU ms':>9s} {'GPU ms':>8s} {'ratio':>7s}")
for name, kind, tc, tg in rows:
    print(f"{name:48s} {kind:62s} {tc:9.3f} {tg:8.4f} {tc / tg:7.0f}x")
