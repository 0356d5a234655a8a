"""dev tool: the shader clock UNDER the Winograd forward inside the training step.  Needs a -DDBEV_WINO_ABLATE build through DBEV_HIP_LIB
and DBEV_WINO_PERSIST=1 (the stamps live in wino_fwdp): workgroup 0 of every launch adds its lifetime in shader cycles (s_memtime) and
in 100 MHz ticks (s_memrealtime); the ratio over all launches of N steps is the clock the kernel actually gets."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd import _lib as L
from distill_bev_amd.train_step import Trainer, build_model, make_batch
dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
h = ctypes.CDLL(L.LIB_PATH)
a = (ctypes.c_ulonglong * 16)(); b = (ctypes.c_ulonglong * 16)()
h.dbev_wino_prof_read(a)
n = 6
for _ in range(n):
    tr.step(batch)
torch.cuda.synchronize()
h.dbev_wino_prof_read(b)
cyc, ticks, items, launches = (int(b[i]) - int(a[i]) for i in (8, 9, 10, 11))
print("wino_fwdp in %d steps: %d launches, %d items of workgroup 0, %.1f ms of workgroup-0 lifetime per step" % (n, launches, items, ticks / 100e3 / n))
print("shader clock under the kernel: %.3f GHz (%d cycles / %d ticks of 100 MHz)" % (cyc / max(ticks, 1) * 0.1, cyc, ticks))
