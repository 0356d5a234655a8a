#!/usr/bin/env python3
"""lift-splat prepare only (dev tool for A/B builds selected with DBEV_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distill_bev_amd import _lib as L, lss as LSS, synthetic as syn
from distill_bev_amd.lift_splat import lift_splat_prepare

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B = 8
rig = {k: torch.from_numpy(v).to(dev) for k, v in syn.camera_rig(B, rng).items()}
dx, bx, nx = LSS.gen_dx_bx([-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8], [-10.0, 10.0, 20.0])
fr = LSS.create_frustum().to(dev)
geom = LSS.get_geometry(fr, rig["rots"], rig["trans"], rig["intrins"], rig["post_rots"], rig["post_trans"]).contiguous()
L.enable_timing("dbev_lift_splat_prepare")
for it in range(30):
    prep = lift_splat_prepare(geom, dx.tolist(), bx.tolist(), [128, 128, 1])
ms = L.timing_ms("dbev_lift_splat_prepare")[5:]
cs = prep.cell_start.cpu().numpy()
ln = np.diff(cs)
print(f"{os.environ.get('DBEV_HIP_LIB','default')[-24:]:>24s} prepare avg {1e3*sum(ms)/len(ms):8.1f} us  min {1e3*min(ms):8.1f} us")
print("   cells", ln.size, "occupied", int((ln > 0).sum()), "<=16:", int(((ln > 0) & (ln <= 16)).sum()), "17..64:", int(((ln > 16) & (ln <= 64)).sum()),
      "65..128:", int(((ln > 64) & (ln <= 128)).sum()), ">128:", int((ln > 128).sum()), "max", int(ln.max()),
      "pts in >64:", int(ln[ln > 64].sum()), "of", int(ln.sum()))
