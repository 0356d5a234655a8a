"""dev tool: where does a work item of the persistent Winograd forward (wino_fwdp) spend its time?  Needs a -DDBEV_WINO_ABLATE build
(tools/build_variant.sh abl -DDBEV_WINO_ABLATE; DBEV_HIP_LIB=distill_bev_amd/libdbev_hip_abl.so): shader-clock stamps of workgroup 0."""
import ctypes, os, sys, torch
sys.path.insert(0, os.getcwd())
from distill_bev_amd import wino, _lib as L
dev = torch.device("cuda:0")
names = ["wait first stages", "first transform", "first k group", "other k groups", "bias + next item", "next geometry + requests", "output transform + stores"]
h = ctypes.CDLL(L.LIB_PATH)
for (N, C, Co, H, W) in [(48, 64, 64, 64, 176), (48, 256, 256, 16, 44), (8, 64, 64, 128, 128), (8, 256, 256, 64, 64), (8, 512, 512, 64, 64)]:
    x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
    U = wino.pack_filters(w, False, x.shape)
    for _ in range(3):
        wino.conv_packed(x, U, Co)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    assert h.dbev_wino_prof_read(out) == 0
    n = max(int(out[7]), 1)
    tot = sum(int(out[i]) for i in range(7))
    print((N, C, Co, H, W), "items of workgroup 0:", n, " cycles per item: %.0f (MFMA alone: %d)" % (tot / n, C // 8 * 4096))
    for i, nm in enumerate(names):
        print("   %-28s %8.0f" % (nm, int(out[i]) / n))
