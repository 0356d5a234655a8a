"""dev tool: forward Winograd kernel time per shape under DBEV_WINO_DBG ablation bits (needs a -DDBEV_WINO_ABLATE build through
DBEV_HIP_LIB).  One process per setting (the bits are read once).   python tools/kbench_wino_abl.py [wg]"""
import os, subprocess, sys
SHAPES = [(48, 256, 256, 16, 44), (8, 256, 256, 64, 64), (48, 64, 64, 64, 176), (8, 512, 512, 64, 64)]
CHILD = r"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from distill_bev_amd import wino
wg = sys.argv[1] == "wg"
dev = torch.device("cuda:0")
out = []
for (N, C, Co, H, W) in %r:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    gy = torch.randn((N, Co, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    U = wino.pack_filters(w)
    f = (lambda: wino.weight_gradient(x, gy, w)) if wg else (lambda: wino.conv_packed(x, U, Co))
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print(" ".join("%%8.1f" %% v for v in out))
""" % (SHAPES,)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
bits = [0, 4, 1, 1024, 512, 8 | 16, 4 | 1024, 4 | 1 | 1024] if mode != "wg" else [0, 64, 128, 64 | 128, 32, 2 << 9, 4 << 9, 1 << 9, 6 << 9]
print("shapes:", SHAPES)
for b in bits:
    env = dict(os.environ, DBEV_WINO_DBG=str(b), DBEV_WINO_HYBRID="0")
    r = subprocess.run([sys.executable, "-c", CHILD, mode], env=env, capture_output=True, text=True)
    print("dbg %5d: %s" % (b, r.stdout.strip() or r.stderr.strip()[-300:]))
