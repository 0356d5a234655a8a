"""weight-gradient entry of one layer shape, timed with events (dev tool; DBEV_WINO_DBG ablation bits are read once per process).
python tools/kbench_wgrad_dbg.py N C Co H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distill_bev_amd import wino
N, C, Co, H, W = [int(v) for v in sys.argv[1:6]]
dev = torch.device("cuda:0")
x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(5):
    wino.weight_gradient(x, gy, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    wino.weight_gradient(x, gy, w)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 30
gf = 32.0 * N * (H // 2) * (W // 2) * C * Co
print("%s dbg=%s  %8.1f us  %6.1f TF (Winograd domain)" % (sys.argv[1:6], os.environ.get("DBEV_WINO_DBG", "0"), us, gf / us / 1e6))
if int(os.environ.get("DBEV_WINO_DBG", "0")) & 256:
    from distill_bev_amd import _lib as L
    nbytes = int(L.call("dbev_wino_conv3x3_backward_weight_workspace_bytes", N, H, W, C, Co))
    ws = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
    gw = torch.empty_like(w)
    L.call("dbev_wino_conv3x3_backward_weight", L.ptr(x), L.ptr(gy), L.ptr(gw), *gw.stride(), N, H, W, C, Co, L.ptr(ws), nbytes, L.stream_ptr(dev))
    torch.cuda.synchronize()
    nsplit = nbytes // (16 * C * Co * 4)
    o = ws.view(torch.int32)[(nsplit - 1) * 16 * C * Co:][:3].tolist()
    print("   clocks %d, 100MHz ticks %d -> %.3f GHz; %d stages: %.0f clocks / stage, %.2f us / stage" % (o[0], o[1], o[0] / (o[1] * 10.0), o[2], o[0] / o[2], o[1] / 100.0 / o[2]))
