#!/usr/bin/env python3
"""Small target for rocprofv3 --pmc passes: a few launches of the streaming kernels at the
training-step shapes (pillars scatter, bev_pool fwd, masked MSE fwd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd import synthetic as syn
from distill_bev_amd.distill_loss import masked_mse_sums
from distill_bev_amd.pillars import pillars_scatter
from distill_bev_amd.voxel import voxelization, dynamic_scatter_prepare

dev = torch.device("cuda:0")
rng = np.random.default_rng(1234)
B = 8
pts = [torch.from_numpy(syn.lidar_points(240000, rng)).to(dev) for _ in range(B)]
vs, rg = [0.2, 0.2, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
co = torch.cat([torch.nn.functional.pad(voxelization(p, vs, rg, -1, -1), (1, 0), value=i) for i, p in enumerate(pts)])
zf = torch.where((co[:, 1:] < 0).any(1), torch.full_like(co[:, 0], -1), co[:, 0])
pr = dynamic_scatter_prepare(torch.stack([zf, co[:, 2], co[:, 3]], 1).contiguous(), grid=(B, 512, 512))
oc = pr["out_coors"]
vc = torch.stack([oc[:, 0], torch.zeros_like(oc[:, 0]), oc[:, 1], oc[:, 2]], 1).int()
vf = torch.randn((pr["M"], 64), device=dev)
print("pillars M =", pr["M"], "algorithmic bytes per launch =", pr["M"] * (4 * 64 + 16) + 4 * 64 * 512 * 512 * B)
for _ in range(5):
    pillars_scatter(vf, vc, B, 512, 512)
for _ in range(5):
    pillars_scatter(vf, vc, B, 512, 512, True)          # NHWC canvas (the bench configuration)
# fused BN(+add)+ReLU at the ResNet-50 layer1 tail shape of the step (48 x 256 x 64 x 176, 554 MB per tensor)
import torch.nn as nn
from distill_bev_amd.bn_act import bn_act
bn = nn.BatchNorm2d(256).to(dev).train()
xb = torch.randn((48, 256, 64, 176), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
rb = torch.randn_like(xb).requires_grad_(True)
for _ in range(3):
    yb = bn_act(xb, bn, rb, True)
    yb.backward(torch.ones_like(yb))
    xb.grad = None; rb.grad = None; bn.zero_grad(set_to_none=True)
S = torch.randn((B, 384, 128, 128), device=dev); T = torch.randn((B, 384, 128, 128), device=dev)
w = torch.rand((B, 1, 128, 128), device=dev)
for _ in range(5):
    masked_mse_sums(S, T, w, w)
# round 2: fused 1x1 adaptation + masked-MSE MFMA kernel at the head position (8 x 256 -> 384 x 128 x 128)
from distill_bev_amd.distill_loss import _FusedAdaptMSE
conv = nn.Conv2d(256, 384, 1).to(dev)
xa = torch.randn((B, 256, 128, 128), device=dev).contiguous(memory_format=torch.channels_last)
Tn = T.contiguous(memory_format=torch.channels_last)
cc = torch.rand((B, 384), device=dev)
with torch.no_grad():
    for _ in range(5):
        _FusedAdaptMSE.apply(xa, conv.weight, conv.bias, Tn, cc)
# round 2: bev_pool surface at the bench shape (16 six-camera frames)
from distill_bev_amd import lss as LSS
from distill_bev_amd.bev_pool import bev_pool
rig = {k: torch.from_numpy(v) for k, v in syn.camera_rig(16, np.random.default_rng(1234)).items()}
dx, bx, nx = LSS.gen_dx_bx([-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8], [-10.0, 10.0, 20.0])
geom = LSS.get_geometry(LSS.create_frustum(), rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"])
coords, _ = LSS.voxel_coords_torch(geom, dx, bx, nx)
coords = coords.to(dev)
fe = torch.randn((coords.shape[0], 64), device=dev).requires_grad_(True)
go = torch.randn((16, 64, 1, 128, 128), device=dev)
print("bev_pool n =", coords.shape[0])
for _ in range(3):
    fe.grad = None
    bev_pool(fe, coords, 16, 1, 128, 128).backward(go)
# round 2: multi-scale deformable attention at BEVFormer's spatial cross-attention geometry (6 cams, 4 FPN levels, 8 points)
from distill_bev_amd.msda import multi_scale_deformable_attn
shapes = torch.tensor([[116, 200], [58, 100], [29, 50], [15, 25]], device=dev)
starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
Sk = int((shapes[:, 0] * shapes[:, 1]).sum())
val = torch.randn((6, Sk, 8, 32), device=dev).requires_grad_(True)
loc = torch.rand((6, 10000, 8, 4, 8, 2), device=dev).requires_grad_(True)
att = torch.softmax(torch.randn((6, 10000, 8, 32), device=dev), -1).view(6, 10000, 8, 4, 8).requires_grad_(True)
for _ in range(3):
    o = multi_scale_deformable_attn(val, shapes, starts, loc, att)
    o.backward(torch.ones_like(o))
    val.grad = None; loc.grad = None; att.grad = None
# round 3: 1x1 convolution + BatchNorm statistics GEMM at the first-stage shape (48 x 64 -> 256 x 64 x 176; 692 MB algorithmic)
from distill_bev_amd import _lib as L
Mc, Kc, Nc = 48 * 64 * 176, 64, 256
xc = torch.randn((Mc, Kc), device=dev); wc = torch.randn((Nc, Kc), device=dev) * 0.1
yc = torch.empty((Mc, Nc), device=dev)
pc = torch.empty((int(L.call("dbev_conv1x1_stats_rows", Mc, Kc, Nc)), 2, Nc), device=dev)
for _ in range(5):
    L.call("dbev_conv1x1_forward", L.ptr(xc), L.ptr(wc), L.ptr(yc), L.ptr(pc), Mc, Kc, Nc, Kc, L.stream_ptr(dev))
# round 3: the 36 final convolutions of the student head in one launch each way (8 x 36 x 64 x 128 x 128 wide map: 1.2 GB)
from distill_bev_amd.head_batch import _BranchFinalConvs
nbr, Chh = 36, 64
Ah = torch.randn((B, nbr * Chh, 128, 128), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
couts = [2, 1, 3, 2, 2, 1] * 6
wbs = []
for co in couts:
    wbs += [(torch.randn((co, Chh, 3, 3), device=dev) * 0.05).requires_grad_(True), torch.zeros((co,), device=dev).requires_grad_(True)]
for _ in range(3):
    ys = _BranchFinalConvs.apply(Ah, Chh, *wbs)
    torch.autograd.grad(sum(y.sum() for y in ys), [Ah] + wbs)
print("head branches: wide map", Ah.numel() * 4, "bytes")
# round 3: stem max pooling at the step's shape (48 x 64 x 128 x 352 -> 64 x 176)
from distill_bev_amd.pool import max_pool
xs = torch.relu(torch.randn((48, 64, 128, 352), device=dev)).contiguous(memory_format=torch.channels_last).requires_grad_(True)
mp = nn.MaxPool2d(3, 2, 1)
for _ in range(3):
    ym = max_pool(mp, xs)
    torch.autograd.grad(ym.sum(), xs)
# round 3: depth-head tail at the step's shape (48 x 256 x 16 x 44 -> 59 depth bins)
from distill_bev_amd.depth_head import depth_head
bnd, cvd = nn.BatchNorm2d(256).to(dev).train(), nn.Conv2d(256, 59, 1).to(dev)
xd = torch.randn((48, 256, 16, 44), device=dev).contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(5):
        depth_head(xd, bnd, cvd)
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (MI355X_MICROARCH.md, HBM section)
buf = torch.empty((128 * 1024 * 1024,), device=dev); src = torch.randn_like(buf)   # 512 MiB each
for _ in range(3):
    buf.copy_(src)      # reads 512 MiB, writes 512 MiB  (kernel: direct_copy / elementwise)
    buf.zero_()         # writes 512 MiB (FillFunctor)
torch.cuda.synchronize()
