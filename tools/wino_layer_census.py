"""which layer shapes the step sends through the Winograd kernels (forward / data gradient: wino.conv_packed, weight gradient:
wino.weight_gradient), with counts per step -- the list the per-layer tables of profiles/r04_wino_vs_miopen.txt should cover.  dev tool."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distill_bev_amd.miopen_tuning import use_shipped_db
use_shipped_db()
from distill_bev_amd.train_step import Trainer, build_model, make_batch
from distill_bev_amd import wino
dev = torch.device("cuda:0")
model, cfg = build_model(allow_synthetic_teacher=True)
tr = Trainer(model, cfg, dev, channels_last=True)
batch = make_batch(8, np.random.default_rng(0), dev, n_points=240000)
tr.step(batch)
cnt = collections.Counter()
o1, o2 = wino.conv_packed, wino.weight_gradient
def c1(x, packed, Cout, *a, **k):
    cnt[("fwd/dgrad", tuple(x.shape), Cout)] += 1
    return o1(x, packed, Cout, *a, **k)
def c2(x, gy, w):
    cnt[("wgrad", tuple(x.shape), w.shape[0])] += 1
    return o2(x, gy, w)
wino.conv_packed, wino.weight_gradient = c1, c2
tr.step(batch)
torch.cuda.synchronize()
for (kind, shp, co), n in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%-10s %3d x  x%s -> %d" % (kind, n, list(shp), co))
