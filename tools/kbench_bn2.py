#!/usr/bin/env python3
"""per-kernel timings of the fused BN entry points at two ResNet-50 shapes (dev tool, A/B builds via DBEV_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from distill_bev_amd import _lib as L
from distill_bev_amd import bn_act as BA
dev = torch.device("cuda:0")
SHAPES = [(48, 256, 64, 176, True), (48, 64, 64, 176, False), (48, 512, 32, 88, True), (48, 128, 32, 88, False),
          (48, 1024, 16, 44, True), (48, 256, 16, 44, False), (48, 2048, 8, 22, True), (8, 128, 128, 128, False),
          (8, 64, 256, 256, False)]
tag = "tpb%s unr%s nbx%s" % tuple(os.environ.get(k, "-") for k in ("DBEV_BN_RTPB", "DBEV_BN_RUNR", "DBEV_BN_RNBX"))
tot_f = tot_b = 0.0
for (N, C, H, W, res) in SHAPES:
    bn = nn.BatchNorm2d(C).to(dev).train()
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(x).requires_grad_(True) if res else None
    g = torch.randn_like(x)
    for k in ("dbev_bn_act_train_forward", "dbev_bn_act_backward"):
        L.enable_timing(k)
    for _ in range(12):
        y = BA.bn_act(x, bn, r, True); y.backward(g)
        x.grad = None; bn.zero_grad(set_to_none=True)
        if r is not None: r.grad = None
    mb = x.numel() * 4 / 1e6
    f = L.timing_ms("dbev_bn_act_train_forward")[4:]; b = L.timing_ms("dbev_bn_act_backward")[4:]
    L.disable_timing()
    pf, pb = (4 if res else 3), (8 if res else 5)
    tot_f += 1e3*sum(f)/len(f); tot_b += 1e3*sum(b)/len(b)
    print(f"{tag:>22s} C={C:4d} {mb:6.0f} MB res={int(res)}: fwd {1e3*sum(f)/len(f):7.1f} us ({pf*mb/(1e3*sum(f)/len(f)):4.2f} TB/s)  bwd {1e3*sum(b)/len(b):7.1f} us ({pb*mb/(1e3*sum(b)/len(b)):4.2f} TB/s)")
print(f"{tag:>22s} TOTAL fwd {tot_f:7.1f} us  bwd {tot_b:7.1f} us")
