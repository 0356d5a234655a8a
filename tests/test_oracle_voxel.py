"""CPU: oracle/voxel.c against (a) the fixtures made from the reference's own
voxelization_cpu.cpp, (b) the compiled reference itself when oracle/_ref is present,
(c) torch.unique-based restatement + brute force for dynamic_scatter (no reference CPU path)."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from distill_bev_amd import synthetic as syn
from oracle import voxel as V


def test_dynamic_and_hard_voxelize_small_fixture():
    g = load_golden("voxel_small.npz")
    vs, rg = g["voxel_size"].tolist(), g["coors_range"].tolist()
    assert np.array_equal(V.dynamic_voxelize(g["points"], vs, rg), g["dyn_coors"])
    v, c, n = V.hard_voxelize(g["points"], vs, rg, int(g["hard5_max_points"]), int(g["hard5_max_voxels"]))
    assert v.shape[0] == int(g["hard5_max_voxels"])       # overflowed: clipped at max_voxels
    assert np.array_equal(c, g["hard5_coors"]) and np.array_equal(n, g["hard5_num"])
    assert np.array_equal(v, g["hard5_voxels"])
    assert n.max() == int(g["hard5_max_points"]) and (n < 2).any()                                    # max_points overflow exercised
    v, c, n = V.hard_voxelize(g["points"], vs, rg, 20, 30000)
    assert np.array_equal(c, g["hard20_coors"]) and np.array_equal(n, g["hard20_num"])
    assert np.array_equal(v, g["hard20_voxels"])
    # planted border rows (make_golden.make_voxel)
    d = g["dyn_coors"]
    assert d[0].tolist() == [0, 0, 0] and d[1].tolist() == [-1, -1, -1]
    assert d[5].tolist() == [-1, -1, -1] and d[6].tolist()[0] == 0


def test_voxelize_3d_grid_fixture():
    g = load_golden("voxel_3d.npz")
    vs, rg = g["voxel_size"].tolist(), g["coors_range"].tolist()
    assert np.array_equal(V.dynamic_voxelize(g["points"], vs, rg), g["dyn_coors"])
    v, c, n = V.hard_voxelize(g["points"], vs, rg, int(g["max_points"]), int(g["max_voxels"]))
    assert np.array_equal(v, g["hard_voxels"]) and np.array_equal(c, g["hard_coors"])
    assert np.array_equal(n, g["hard_num"])


def test_voxelize_30k_hashes():
    g = load_golden("voxel_30k_stats.npz")
    pts = syn.lidar_points(30000, np.random.default_rng(1234))
    vs, rg = [0.2, 0.2, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    d = V.dynamic_voxelize(pts, vs, rg)
    assert hashlib.sha256(d.tobytes()).digest() == g["dyn_sha256"].tobytes()
    assert int((d[:, 0] < 0).sum()) == int(g["n_invalid"])
    v, c, n = V.hard_voxelize(pts, vs, rg, 20, 30000)
    assert v.shape[0] == int(g["n_voxels"])
    assert hashlib.sha256(c.tobytes()).digest() == g["hard_coors_sha256"].tobytes()
    assert hashlib.sha256(n.tobytes()).digest() == g["hard_num_sha256"].tobytes()
    assert hashlib.sha256(v.tobytes()).digest() == g["hard_voxels_sha256"].tobytes()


def test_against_compiled_reference_when_present():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    ref = build_ref.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref not built on this box")
    rng = np.random.default_rng(99)
    for trial in range(3):
        pts = syn.lidar_points(5000, rng)
        vs, rg = [0.2, 0.2, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
        t = torch.from_numpy(pts)
        coors = t.new_zeros((t.size(0), 3), dtype=torch.int)
        ref.dynamic_voxelize(t, coors, vs, rg, 3)
        assert np.array_equal(coors.numpy(), V.dynamic_voxelize(pts, vs, rg))
        mv, mp = 3000 + 500 * trial, 3 + trial
        voxels = t.new_zeros((mv, mp, 5)); c = t.new_zeros((mv, 3), dtype=torch.int)
        n = t.new_zeros((mv,), dtype=torch.int)
        m = ref.hard_voxelize(t, voxels, c, n, vs, rg, mp, mv, 3, True)
        v2, c2, n2 = V.hard_voxelize(pts, vs, rg, mp, mv)
        assert m == v2.shape[0]
        assert np.array_equal(voxels[:m].numpy(), v2) and np.array_equal(c[:m].numpy(), c2)
        assert np.array_equal(n[:m].numpy(), n2)


def _torch_scatter_restatement(feats, coors, reduce_type):
    """scatter_points_cuda.cu:199-233 restated with the very torch ops the CUDA host code
    calls (masked_fill, unique_dim sorted) + a dense loop for the reduction."""
    f = torch.from_numpy(feats)
    c = torch.from_numpy(coors)
    clean = c.masked_fill(c.lt(0).any(-1, True), -1)
    out_coors, cmap, cnt = torch.unique(clean, dim=0, sorted=True, return_inverse=True, return_counts=True)
    if out_coors[0, 0] < 0:
        out_coors, cnt, cmap = out_coors[1:], cnt[1:], cmap - 1
    M = out_coors.shape[0]
    red = torch.full((M, f.shape[1]), float("-inf") if reduce_type == "max" else 0.0, dtype=torch.float64)
    for i in range(f.shape[0]):
        to = int(cmap[i])
        if to < 0:
            continue
        if reduce_type == "max":
            red[to] = torch.maximum(red[to], f[i].double())
        else:
            red[to] += f[i].double()
    if reduce_type == "mean":
        red /= cnt.unsqueeze(-1).double()
    return red.float().numpy(), out_coors.numpy().astype(np.int32), cmap.numpy().astype(np.int32), cnt.numpy().astype(np.int32)


@pytest.mark.parametrize("reduce_type", ["max", "mean", "sum"])
def test_dynamic_scatter_forward_matches_torch_unique_restatement(reduce_type):
    rng = np.random.default_rng(3)
    pts = syn.lidar_points(2500, rng)
    coors = V.dynamic_voxelize(pts, [0.8, 0.8, 8.0], [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0])
    coors[7] = [-1, 5, 5]      # partially negative row must be cleaned to (-1,-1,-1)
    feats = rng.normal(size=(2500, 6)).astype(np.float32)
    red, oc, cmap, cnt = V.dynamic_scatter_forward(feats, coors, reduce_type)
    r2, oc2, cmap2, cnt2 = _torch_scatter_restatement(feats, coors, reduce_type)
    assert np.array_equal(oc, oc2) and np.array_equal(cmap, cmap2) and np.array_equal(cnt, cnt2)
    assert cmap[7] == -1
    if reduce_type == "max":
        assert np.array_equal(red, r2)
    else:
        assert np.abs(red - r2).max() < 1e-5
    # lexicographic (z, y, x) order of the output rows
    lin = (oc[:, 0].astype(np.int64) * 10**8 + oc[:, 1] * 10**4 + oc[:, 2])
    assert np.all(np.diff(lin) > 0)


def test_dynamic_scatter_all_valid_and_empty():
    feats = np.arange(12, dtype=np.float32).reshape(4, 3)
    coors = np.array([[0, 1, 1], [0, 0, 2], [0, 1, 1], [0, 0, 2]], dtype=np.int32)
    red, oc, cmap, cnt = V.dynamic_scatter_forward(feats, coors, "max")
    assert oc.tolist() == [[0, 0, 2], [0, 1, 1]] and cmap.tolist() == [1, 0, 1, 0] and cnt.tolist() == [2, 2]
    assert red.tolist() == [[9, 10, 11], [6, 7, 8]]
    red, oc, cmap, cnt = V.dynamic_scatter_forward(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), "max")
    assert red.shape == (0, 3) and cmap.shape == (0,)


@pytest.mark.parametrize("reduce_type", ["max", "mean", "sum"])
def test_dynamic_scatter_backward_matches_autograd_of_restatement(reduce_type):
    rng = np.random.default_rng(4)
    n, c = 400, 5
    coors = rng.integers(0, 6, (n, 3)).astype(np.int32)
    coors[::17] = -1
    feats = rng.normal(size=(n, c)).astype(np.float32)
    feats[10] = feats[3]; coors[10] = coors[3]          # exact tie -> lowest index wins for max
    red, oc, cmap, cnt = V.dynamic_scatter_forward(feats, coors, reduce_type)
    gr = rng.normal(size=red.shape).astype(np.float32)
    g = V.dynamic_scatter_backward(gr, feats, red, cmap, cnt, reduce_type)
    # brute force
    gb = np.zeros_like(feats)
    for v in range(red.shape[0]):
        members = np.flatnonzero(cmap == v)
        for k in range(c):
            if reduce_type == "sum":
                gb[members, k] = gr[v, k]
            elif reduce_type == "mean":
                gb[members, k] = gr[v, k] / np.float32(cnt[v])
            else:
                hit = members[feats[members, k] == red[v, k]]
                gb[hit.min(), k] = gr[v, k]
    assert np.array_equal(g, gb)
    assert np.all(g[cmap < 0] == 0)


def test_pillars_scatter_fixture():
    g = load_golden("pillars_scatter_small.npz")
    canvas = V.pillars_scatter(g["feats"], g["coors"], int(g["B"]), int(g["ny"]), int(g["nx"]))
    assert np.array_equal(canvas, g["canvas"])
