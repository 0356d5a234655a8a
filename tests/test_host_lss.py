"""CPU: host-side LSS helpers (distill_bev_amd/lss.py) and the torch CPU-baseline port
(oracle/lss_torch.py) against the reference fixtures."""
import numpy as np
import torch

from conftest import load_golden
from distill_bev_amd import lss as LSS
from oracle import lss_torch as OT


def test_frustum_and_grid_match_reference():
    g = load_golden("lss_small.npz")
    fr = LSS.create_frustum(tuple(g["input_size"].tolist()), 16, tuple(g["dbound"].tolist()))
    assert np.array_equal(fr.numpy(), g["frustum"])
    dx, bx, nx = LSS.gen_dx_bx(g["xbound"].tolist(), g["ybound"].tolist(), g["zbound"].tolist())
    assert np.array_equal(dx.numpy(), g["dx"]) and np.array_equal(bx.numpy(), g["bx"])
    assert np.array_equal(nx.numpy(), g["nx"])


def test_get_geometry_matches_reference_on_cpu():
    """Same arithmetic as vt_mine.get_geometry (the two broadcast matmuls are written as
    multiply-adds): within a few ulp of the reference fixture (1e-4 m bound, positions <= 60 m)."""
    s = load_golden("lss_small.npz")
    g = load_golden("lss_lift_small.npz")
    t = {k: torch.from_numpy(s[k]) for k in ("rots", "trans", "intrins", "post_rots", "post_trans")}
    geom = LSS.get_geometry(torch.from_numpy(s["frustum"]), **t)
    assert np.abs(geom.numpy() - g["geom"]).max() < 1e-4
    f = load_golden("lss_full_stats.npz")
    t = {k: torch.from_numpy(f[k]) for k in ("rots", "trans", "intrins", "post_rots", "post_trans")}
    geomf = LSS.get_geometry(LSS.create_frustum(), **t)
    assert np.abs(geomf.numpy()[:, :, ::6] - f["geom_full"]).max() < 1e-4
    coords, kept = LSS.voxel_coords_torch(geomf, *LSS.gen_dx_bx([-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8], [-10.0, 10.0, 20.0]))
    assert abs(int(kept.sum()) - int(f["n_kept"])) <= 5      # border points may move with the last ulp


def test_cpu_baseline_port_matches_reference_output():
    g = load_golden("lss_small.npz")
    out = OT.voxel_pooling_cumsum(torch.from_numpy(g["geom"]), torch.from_numpy(g["x"]),
                                  torch.from_numpy(g["dx"]), torch.from_numpy(g["bx"]),
                                  torch.from_numpy(g["nx"]))
    assert np.abs(out.numpy() - g["out"]).max() < 1e-5
    l = load_golden("lss_lift_small.npz")
    vol = OT.lift(torch.from_numpy(l["depth"]), torch.from_numpy(l["img_feat"]), 2, 2)
    bev = OT.voxel_pooling_cumsum(torch.from_numpy(l["geom"]), vol, torch.from_numpy(l["dx"]),
                                  torch.from_numpy(l["bx"]), torch.from_numpy(l["nx"]))
    assert np.abs(bev.numpy() - l["bev"]).max() < 1e-5
