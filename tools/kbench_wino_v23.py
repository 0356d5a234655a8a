"""forward Winograd kernels wino_fwd (DBEV_WINO_FWD_V=2) vs wino_fwd3 (=3) on the step's layer shapes: one process per version (the
choice is read once), this script runs both and prints the table wino_plan's threshold is read from.
python tools/kbench_wino_v23.py            (parent)      python tools/kbench_wino_v23.py child   (one version, env set by the parent)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SH = [(48, 256, 256, 16, 44), (48, 128, 128, 32, 88), (48, 64, 64, 64, 176), (48, 512, 512, 8, 22), (8, 64, 64, 128, 128),
      (8, 512, 256, 128, 128), (8, 128, 128, 128, 128), (8, 256, 256, 64, 64), (8, 512, 512, 64, 64), (8, 640, 512, 64, 64),
      (8, 64, 2304, 128, 128), (8, 64, 64, 256, 256), (8, 256, 256, 32, 32), (8, 512, 512, 16, 16), (48, 512, 512, 16, 44),
      (8, 384, 2304, 128, 128), (8, 128, 128, 64, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from distill_bev_amd import wino
    dev = torch.device("cuda:0")
    for N, C, Co, H, W in SH:
        x = torch.randn((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((Co, C, 3, 3), device=dev) / (3 * C ** 0.5)).contiguous(memory_format=torch.channels_last)
        U = wino.pack_filters(w)
        for _ in range(3):
            wino.conv_packed(x, U, Co)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for _ in range(20):
            wino.conv_packed(x, U, Co)
        b.record(); torch.cuda.synchronize()
        print("%.1f" % (a.elapsed_time(b) / 20 * 1e3), flush=True)
    sys.exit(0)
res = {}
for rep in range(2):
    for v in ("2", "3"):
        out = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DBEV_WINO_FWD_V=v), capture_output=True, text=True).stdout.split()
        res.setdefault(v, []).append([float(t) for t in out[-len(SH):]])
print("%-28s %9s %9s   %s" % ("shape (N, C, Co, H, W)", "fwd us", "fwd3 us", "64-tile items x channel blocks"))
for i, (N, C, Co, H, W) in enumerate(SH):
    t2 = min(r[i] for r in res["2"]); t3 = min(r[i] for r in res["3"])
    TH, TW = H // 2, W // 2
    nb = min(-(-TH // 8) * -(-TW // 8), -(-TH // 4) * -(-TW // 16)) * N * (Co // 64)
    print("%-28s %9.1f %9.1f   %6d  %s" % (str((N, C, Co, H, W)), t2, t3, nb, "fwd3" if t3 < t2 else "fwd"))
