"""Per-layer view of the voxel teacher's sparse convolutions: sites, valid (site, offset) pairs, MFMA work the kernel actually
issues (every offset for which ANY site of a 128-site tile has a neighbour) and the time of each sp_conv_fwd launch."""
import numpy as np
import torch

import bench_workloads as W
from distill_bev_amd import _lib as L
from distill_bev_amd import spconv

dev = torch.device("cuda:0")
w = W.VoxelTeacher(dev, 0, 1)
recs = []


def hook(mod, inp, out):
    if mod.conv1x1:
        return
    x = inp[0]
    key = mod.indice_key if mod.indice_key is not None else mod._auto_key(x)
    rb = x.rulebooks[key]
    nbr = rb.nbr[:rb.n_out]
    valid = int((nbr >= 0).sum())
    t = (nbr >= 0).view(-1, nbr.shape[1])
    pad = (-t.shape[0]) % 128
    if pad:
        t = torch.cat([t, t.new_zeros((pad, t.shape[1]))])
    issued = int(t.view(-1, 128, t.shape[1]).any(1).sum()) * 128
    recs.append([type(mod).__name__, mod.in_channels, mod.out_channels, rb.n_out, rb.K, valid, issued])


hs = [m.register_forward_hook(hook) for m in w.model.modules() if isinstance(m, spconv.SparseConvolution)]
w.step()
for h in hs:
    h.remove()
for _ in range(3):
    w.step()
torch.cuda.synchronize()
L.kernel_timing_read()
L.kernel_timing(["sp_conv_fwd"])
w.step()
rec = L.kernel_timing_read()["sp_conv_fwd"]
L.kernel_timing(False)
assert len(rec) == len(recs), (len(rec), len(recs))
print(f"{'layer':14s} {'cin':>4s} {'cout':>4s} {'sites':>9s} {'K':>3s} {'valid/site':>10s} {'issued/valid':>12s} {'ms':>7s} {'TF valid':>9s} {'TF issued':>9s}")
tot = [0.0, 0.0, 0.0]
for r, (ms, _) in zip(recs, rec):
    name, ci, co, n, K, valid, issued = r
    ci16, co16 = (ci + 15) // 16 * 16, (co + 15) // 16 * 16
    fv, fi = 2.0 * valid * ci * co, 2.0 * issued * ci16 * co16
    tot[0] += ms; tot[1] += fv; tot[2] += fi
    print(f"{name:14s} {ci:4d} {co:4d} {n:9d} {K:3d} {valid / n:10.2f} {issued / max(valid, 1):12.2f} {ms:7.3f} {fv / ms / 1e9:9.1f} {fi / ms / 1e9:9.1f}")
print(f"total {tot[0]:.2f} ms   valid {tot[1] / tot[0] / 1e9:.1f} TFLOP/s   issued {tot[2] / tot[0] / 1e9:.1f} TFLOP/s")
