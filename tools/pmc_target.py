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
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (MI355X_MICROARCH.md, HBM section)
buf = torch.empty((128 * 1024 * 1024,), device=dev); src = torch.randn_like(buf)   # 512 MiB each
for _ in range(3):
    buf.copy_(src)      # reads 512 MiB, writes 512 MiB  (kernel: direct_copy / elementwise)
    buf.zero_()         # writes 512 MiB (FillFunctor)
torch.cuda.synchronize()
