#!/usr/bin/env python3
"""lift-splat kernels only, at the training-step shapes (dev tool; DBEV_HIP_LIB selects an A/B build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd import _lib as L, lss as LSS, synthetic as syn
from distill_bev_amd.lift_splat import lift_splat, lift_splat_prepare

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B = int(os.environ.get("B", 8))
rig = {k: torch.from_numpy(v).to(dev) for k, v in syn.camera_rig(B, rng).items()}
dx, bx, nx = LSS.gen_dx_bx([-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8], [-10.0, 10.0, 20.0])
fr = LSS.create_frustum().to(dev)
geom = LSS.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"]).contiguous()
prep = lift_splat_prepare(geom, dx.tolist(), bx.tolist(), [128, 128, 1])
d, f = syn.lss_inputs(B, rng)
dt = torch.from_numpy(d).to(dev).requires_grad_(True)
ft = torch.from_numpy(f).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
NAMES = ("dbev_lift_splat_forward", "dbev_lift_splat_backward", "dbev_lift_splat_prepare")
for k in NAMES:
    L.enable_timing(k)
for it in range(30):
    prep = lift_splat_prepare(geom, dx.tolist(), bx.tolist(), [128, 128, 1])
    bev = lift_splat(dt, ft, prep)
    g = torch.randn_like(bev)
    torch.autograd.grad(bev, (dt, ft), g)
torch.cuda.synchronize()
for k in NAMES:
    ms = L.timing_ms(k)[5:]
    print(f"{os.environ.get('DBEV_HIP_LIB','default')[-24:]:>24s} {k:28s} avg {1e3*sum(ms)/len(ms):8.1f} us  min {1e3*min(ms):8.1f} us")
