"""one layer shape through the Winograd kernels (forward, weight gradient) a few times -- the target of rocprofv3 --pmc passes.
python tools/kbench_wino_one.py N C Co H W [iters]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distill_bev_amd import wino
N, C, Co, H, W = [int(v) for v in sys.argv[1:6]]
it = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
U = wino.pack_filters(w)
for _ in range(it):
    wino.conv_packed(x, U, Co)
    if C % 64 == 0:
        wino.weight_gradient(x, gy, w)
torch.cuda.synchronize()
