"""forward Winograd kernel on two shapes under the DBEV_WINO_VAR / DBEV_WINO_DBG settings of the environment (one process per setting)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distill_bev_amd import wino
dev = torch.device("cuda:0")
out = []
for N, C, Co, H, W in [(8, 512, 512, 64, 64), (48, 64, 64, 64, 176), (48, 256, 256, 16, 44)]:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    U = wino.pack_filters(w)
    for _ in range(3):
        y = wino.conv_packed(x, U, Co)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(20):
        y = wino.conv_packed(x, U, Co)
    b.record(); torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x[:1], w, None, 1, 1)
    err = float((y[:1] - ref).abs().max() / ref.abs().max())
    out.append(f"{a.elapsed_time(b) / 20 * 1e3:8.1f} us (err {err:.0e})")
print(f"VAR={os.environ.get('DBEV_WINO_VAR', '0')} DBG={os.environ.get('DBEV_WINO_DBG', '0')}: " + " | ".join(out), flush=True)
