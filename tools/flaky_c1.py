"""Repeat a bottleneck block forward + backward through the library path and through the fused 1x1-convolution GEMM: which of the
two is not bit-stable run to run?  (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from distill_bev_amd import bn_act as BA
from distill_bev_amd.nets import Bottleneck
dev = torch.device("cuda:0")
inplanes, planes = 128, 32
torch.manual_seed(3)
blk = Bottleneck(inplanes, planes).to(dev).to(memory_format=torch.channels_last).train()
x0 = torch.randn(4, inplanes, 24, 40, device=dev).contiguous(memory_format=torch.channels_last)
gy = torch.randn(4, planes * 4, 24, 40, device=dev).contiguous(memory_format=torch.channels_last)
sd0 = {k: v.clone() for k, v in blk.state_dict().items()}


def run(mode):
    blk.load_state_dict(sd0)
    BA._C1.update(enabled=mode == "fused", min_rows=1)
    x = x0.clone().requires_grad_(True)
    y = blk(x)
    params = dict(blk.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), gy)
    return [y.detach()] + list(grads)


for mode in ("library", "fused"):
    ref = run(mode)
    bad = {}
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
        out = run(mode)
        for i, (a, b) in enumerate(zip(ref, out)):
            if not torch.equal(a, b):
                e = float((a - b).norm() / a.norm())
                bad.setdefault(i, []).append((it, e))
    print(mode, "tensors that differed (index: count, max rel err):", {i: (len(v), max(e for _, e in v)) for i, v in bad.items()})
names = ["y", "gx"] + ["g_" + n for n, _ in blk.named_parameters()]
print(names)
