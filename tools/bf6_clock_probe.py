"""dev tool: shader clock inside the bf16x6 forward kernel (s_memtime / s_memrealtime of workgroup 0; needs a -DDBEV_BF6_ABLATE build\nthrough DBEV_HIP_LIB and DBEV_BF6_DBG=16)"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from distill_bev_amd import gemm_bf6 as G
dev = torch.device("cuda:0")
for (n, ci, co, h, w) in [(48, 256, 1024, 16, 44), (48, 1024, 256, 16, 44), (8, 512, 512, 64, 64)]:
    x = torch.relu(torch.randn((n, ci, h, w), device=dev)).contiguous(memory_format=torch.channels_last)
    wt = torch.randn((co, ci, 1, 1), device=dev) / ci ** 0.5
    for rep in (1, 30):
        for _ in range(rep):
            y = G.product(x, wt)
        torch.cuda.synchronize()
        v = y.permute(0, 2, 3, 1).reshape(-1)[:2].view(torch.int32).tolist()
        print((n, ci, co, h, w), "after", rep, "launches: shader cycles", v[0], "100MHz ticks", v[1], "-> %.2f GHz" % (v[0] / max(v[1], 1) * 0.1))
