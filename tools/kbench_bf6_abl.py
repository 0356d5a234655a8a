"""dev tool: the bf16x6 GEMM under DBEV_BF6_DBG ablation bits (a -DDBEV_BF6_ABLATE build through DBEV_HIP_LIB), us per launch"""
import os, subprocess, sys
SHAPES = [(48, 256, 1024, 16, 44), (48, 1024, 256, 16, 44), (48, 64, 256, 64, 176), (8, 512, 512, 64, 64)]
CHILD = r"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from distill_bev_amd import gemm_bf6 as G
dev = torch.device("cuda:0")
out = []
for (n, ci, co, h, w) in %r:
    x = torch.relu(torch.randn((n, ci, h, w), device=dev)).contiguous(memory_format=torch.channels_last)
    wt = torch.randn((co, ci, 1, 1), device=dev) / ci ** 0.5
    f = lambda: G.product(x, wt)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20 * 1e3)
print(" ".join("%%8.1f" %% v for v in out))
""" % (SHAPES,)
print("shapes:", SHAPES)
for b in [0, 1, 2, 3, 4, 8, 1 | 2 | 4, 15]:
    env = dict(os.environ, DBEV_BF6_DBG=str(b))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("dbg %3d: %s" % (b, r.stdout.strip() or r.stderr.strip()[-300:]))
