"""one layer shape through the Winograd kernels (forward, weight gradient) a few times -- the target of rocprofv3 --pmc passes.
python tools/kbench_wino_one.py N C Co H W [iters] [spaced] [nowgrad]
spaced: a streaming pass over a 554 MB buffer between the launches, as a normalisation pass follows every convolution in the step (20
identical fp32-MFMA launches back to back run 25-30 % slower than the same launch inside the step: the board's power limit)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distill_bev_amd import wino
N, C, Co, H, W = [int(v) for v in sys.argv[1:6]]
it = int(sys.argv[6]) if len(sys.argv) > 6 else 5
spaced = "spaced" in sys.argv[7:]
nowgrad = "nowgrad" in sys.argv[7:]
dev = torch.device("cuda:0")
x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
big = torch.randn((138 * 1024 * 1024,), device=dev) if spaced else None
U = wino.pack_filters(w, False, x.shape)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * it)]
for i in range(it):
    if spaced:
        big.mul_(1.0001)
    ev[2 * i].record()
    wino.conv_packed(x, U, Co)
    ev[2 * i + 1].record()
    if C % 64 == 0 and not nowgrad:
        wino.weight_gradient(x, gy, w)
torch.cuda.synchronize()
ts = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) * 1e3 for i in range(it))
fl = 32 * N * (H // 2) * (W // 2) * C * Co
print("shape", (N, C, Co, H, W), "spaced" if spaced else "back-to-back", "fwd us median %.1f min %.1f  -> %.1f TF winograd-domain = %.3f of 157.3"
      % (ts[len(ts) // 2], ts[0], fl / ts[len(ts) // 2] / 1e6, fl / ts[len(ts) // 2] / 1e6 / 157.3))
