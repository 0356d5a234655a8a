"""one layer shape through the bf16x6 GEMM a few times (the target of rocprofv3 --pmc passes).  python tools/kbench_bf6_one.py N Cin Cout H W [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd import gemm_bf6 as G
n, ci, co, h, w = [int(v) for v in sys.argv[1:6]]
it = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
x = torch.relu(torch.randn((n, ci, h, w), device=dev)).contiguous(memory_format=torch.channels_last)
wt = (torch.randn((co, ci, 1, 1), device=dev) / ci ** 0.5)
for _ in range(it):
    G.product(x, wt)
torch.cuda.synchronize()
