#!/usr/bin/env python3
"""Micro-benchmarks of the hand-written kernels at the training-step shapes (dev tool):
event-timed average over N launches + algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd import _lib as L, lss as LSS, synthetic as syn
from distill_bev_amd.lift_splat import lift_splat, lift_splat_prepare
from distill_bev_amd.pillars import pillars_scatter
from distill_bev_amd.voxel import voxelization, dynamic_scatter_prepare, dynamic_scatter_reduce
from distill_bev_amd.distill_loss import abs_mean_maps, masked_mse_sums

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


buf = torch.empty((64 * 512 * 512 * 8,), device=dev); src = torch.randn_like(buf)
t = timeit(lambda: buf.zero_())
print(f"[probe] torch zero_ 537 MB: {t:8.1f} us  {buf.numel()*4/t/1e3:7.1f} GB/s written")
t = timeit(lambda: buf.copy_(src))
print(f"[probe] torch copy_ 537 MB: {t:8.1f} us  {2*buf.numel()*4/t/1e3:7.1f} GB/s read+written")
del buf, src
B = 8
rig = {k: torch.from_numpy(v).to(dev) for k, v in syn.camera_rig(B, rng).items()}
dx, bx, nx = LSS.gen_dx_bx([-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8], [-10.0, 10.0, 20.0])
fr = LSS.create_frustum().to(dev)
geom = LSS.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"]).contiguous()
t = timeit(lambda: lift_splat_prepare(geom, dx.tolist(), bx.tolist(), [128, 128, 1]))
print(f"lift_splat_prepare  B={B}: {t:8.1f} us")
prep = lift_splat_prepare(geom, dx.tolist(), bx.tolist(), [128, 128, 1])
print("   n_kept", int(prep.n_kept), "n_hot", int(prep.n_hot))
d, f = syn.lss_inputs(B, rng)
dt = torch.from_numpy(d).to(dev).requires_grad_(True)
ft = torch.from_numpy(f).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
t = timeit(lambda: lift_splat(dt, ft, prep))
alg = B * (4 * 6 * 59 * 16 * 44 + 4 * 6 * 64 * 16 * 44 + 4 * 128 * 128 * 64)
print(f"lift_splat fwd      B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic ({alg/1e6:.1f} MB)")
bev = lift_splat(dt, ft, prep)
g = torch.randn_like(bev)
t = timeit(lambda: torch.autograd.grad(bev, (dt, ft), g, retain_graph=True))
alg = B * (4 * 128 * 128 * 64 + 2 * (4 * 6 * 59 * 16 * 44 + 4 * 6 * 64 * 16 * 44))
print(f"lift_splat bwd      B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic ({alg/1e6:.1f} MB)")

pts = [torch.from_numpy(syn.lidar_points(240000, rng)).to(dev) for _ in range(B)]
vs, rg = [0.2, 0.2, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
t = timeit(lambda: [voxelization(p, vs, rg, -1, -1) for p in pts])
print(f"dynamic_voxelize    {B}x240k: {t:8.1f} us   {B*240000*32 / t / 1e3:7.1f} GB/s")
co = torch.cat([torch.nn.functional.pad(voxelization(p, vs, rg, -1, -1), (1, 0), value=i) for i, p in enumerate(pts)])
zf = torch.where((co[:, 1:] < 0).any(1), torch.full_like(co[:, 0], -1), co[:, 0])
folded = torch.stack([zf, co[:, 2], co[:, 3]], 1).contiguous()
t = timeit(lambda: dynamic_scatter_prepare(folded, grid=(B, 512, 512)), n=10)
print(f"dyn_scatter_prepare N={folded.shape[0]}: {t:8.1f} us (includes 1 host sync)")
pr = dynamic_scatter_prepare(folded, grid=(B, 512, 512))
feats = torch.randn((folded.shape[0], 64), device=dev)
t = timeit(lambda: dynamic_scatter_reduce(feats, pr, "max"))
alg = folded.shape[0] * (4 * 64 + 4) + pr["M"] * (4 * 64 + 12)
print(f"dyn_scatter_reduce max C=64 M={pr['M']}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic")
vf = dynamic_scatter_reduce(feats, pr, "max")
oc = pr["out_coors"]
vc = torch.stack([oc[:, 0], torch.zeros_like(oc[:, 0]), oc[:, 1], oc[:, 2]], 1).int()
t = timeit(lambda: pillars_scatter(vf, vc, B, 512, 512))
alg = pr["M"] * (4 * 64 + 16) + 4 * 64 * 512 * 512 * B
print(f"pillars_scatter NCHW B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic ({alg/1e6:.0f} MB)")
t = timeit(lambda: pillars_scatter(vf, vc, B, 512, 512, True))
print(f"pillars_scatter NHWC B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic")

S = torch.randn((B, 384, 128, 128), device=dev).requires_grad_(True); T = torch.randn((B, 384, 128, 128), device=dev)
w1 = torch.rand((B, 1, 128, 128), device=dev); w2 = torch.rand((B, 1, 128, 128), device=dev); w3 = torch.rand((B, 1, 128, 128), device=dev)
cc = torch.rand((B, 384), device=dev)
t = timeit(lambda: masked_mse_sums(S, T, w1, w2, w3, cc))
alg = B * (4 * 128 * 128 * 768 + 12 * 128 * 128)
print(f"masked_mse fwd head B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic ({alg/1e6:.0f} MB)")
out = masked_mse_sums(S, T, w1, w2, w3, cc)
t = timeit(lambda: torch.autograd.grad(out.sum(), S, retain_graph=True))
alg = B * (4 * 128 * 128 * 384 * 3)
print(f"masked_mse bwd head B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic")
t = timeit(lambda: abs_mean_maps(T))
alg = B * 4 * 128 * 128 * 384
print(f"abs_mean_maps head  B={B}: {t:8.1f} us   {alg / t / 1e3:7.1f} GB/s algorithmic")

from distill_bev_amd.distill_loss import UpsampleBilinearAC
for cl in (False, True):
    x = torch.randn((B, 256, 32, 32), device=dev)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    up_t = torch.nn.Upsample(scale_factor=4, mode="bilinear", align_corners=True)
    up_m = UpsampleBilinearAC(4)
    yt = up_t(x); ym = up_m(x); g = torch.randn_like(ym)
    t1 = timeit(lambda: up_t(x)); t2 = timeit(lambda: up_m(x))
    t3 = timeit(lambda: torch.autograd.grad(yt, x, g, retain_graph=True)); t4 = timeit(lambda: torch.autograd.grad(ym, x, g, retain_graph=True))
    mb = B * 256 * 128 * 128 * 4 / 1e6
    print(f"upsample x4 8x256x32x32 cl={cl}: fwd torch {t1:7.1f} us | hip {t2:7.1f} us ({mb/t2*1e3:6.0f} GB/s)   bwd torch {t3:7.1f} us | hip {t4:7.1f} us")

# fused teacher pillar path (dbev_pillar_vfe_canvas) at the step shapes
from distill_bev_amd.config import Config
from distill_bev_amd.registry import build_detector
from distill_bev_amd.train_step import DEFAULT_CONFIG
import distill_bev_amd.detectors  # noqa
from distill_bev_amd import pillar_encoder as PE
teacher = build_detector(Config.fromfile(DEFAULT_CONFIG).teacher["model"]).to(dev).eval()
t = timeit(lambda: PE.fused_pillar_canvas(pts, teacher.pts_voxel_layer, teacher.pts_voxel_encoder, teacher.pts_middle_encoder), n=10)
print(f"fused teacher pillar path (8 x 240k pts -> canvas): {t:8.1f} us")
