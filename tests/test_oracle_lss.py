"""CPU: the LSS oracle (oracle/lss.py) against the fixtures produced by the
imported reference view_transformer_mine.py (tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import torch

from conftest import load_golden
from oracle import lss as O


def test_gen_dx_bx_and_frustum_small():
    g = load_golden("lss_small.npz")
    dx, bx, nx = O.gen_dx_bx(g["xbound"].tolist(), g["ybound"].tolist(), g["zbound"].tolist())
    assert np.array_equal(dx, g["dx"]) and np.array_equal(bx, g["bx"]) and np.array_equal(nx, g["nx"])
    fr = O.create_frustum(tuple(g["input_size"].tolist()), 16, tuple(g["dbound"].tolist()))
    assert fr.dtype == np.float32 and np.array_equal(fr, g["frustum"])


def test_frustum_full_bit_exact():
    g = load_golden("lss_full_stats.npz")
    fr = O.create_frustum()
    assert fr.shape == (59, 16, 44, 3)
    assert hashlib.sha256(fr.tobytes()).digest() == g["frustum_sha256"].tobytes()
    # and against torch.linspace on this box
    assert np.array_equal(O.torch_linspace_f32(0, 703, 44), torch.linspace(0, 703, 44).numpy())
    assert np.array_equal(O.torch_linspace_f32(0, 255, 16), torch.linspace(0, 255, 16).numpy())


def test_geometry_small_tolerance():
    g = load_golden("lss_lift_small.npz")
    s = load_golden("lss_small.npz")
    geom = O.get_geometry(s["frustum"], s["rots"], s["trans"], s["intrins"], s["post_rots"], s["post_trans"])
    assert geom.shape == g["geom"].shape
    # tolerance: 1e-4 m (fp32 inverse via LU in torch vs fp64->fp32 here)
    assert np.abs(geom - g["geom"]).max() < 1e-4


def test_geometry_full_tolerance():
    g = load_golden("lss_full_stats.npz")
    geom = O.get_geometry(O.create_frustum(), g["rots"], g["trans"], g["intrins"], g["post_rots"], g["post_trans"])
    assert np.abs(geom[:, :, ::6] - g["geom_full"]).max() < 1e-4
    assert np.abs(geom.reshape(-1, 3)[::997] - g["geom_sample"]).max() < 1e-4


def test_voxel_index_bit_exact_small_incl_trunc_cases():
    g = load_golden("lss_small.npz")
    idx, kept = O.voxel_index(g["geom"], g["dx"], g["bx"], g["nx"])
    assert np.array_equal(idx, g["idx"])
    flat = idx.reshape(-1, 3)
    # adversarial rows planted by make_golden: trunc-toward-zero semantics
    assert flat[0].tolist() == [0, 0, 0]
    assert flat[1][0] == 0      # x in (-1, 0) cell is truncated to 0 -> kept
    assert flat[2][0] == -1 and not kept.reshape(-1)[2]
    assert flat[3][0] == 8 and not kept.reshape(-1)[3]
    assert flat[5][2] == 0 and kept.reshape(-1)[5]


def test_voxel_index_bit_exact_full_sample():
    g = load_golden("lss_full_stats.npz")
    dx, bx, nx = O.gen_dx_bx([-51.2, 51.2, 0.8], [-51.2, 51.2, 0.8], [-10.0, 10.0, 20.0])
    assert np.array_equal(dx, g["dx"]) and np.array_equal(bx, g["bx"]) and np.array_equal(nx, g["nx"])
    idx, _ = O.voxel_index(g["geom_sample"], dx, bx, nx)
    assert np.array_equal(idx, g["idx_sample"])


def test_voxel_pooling_small_vs_reference():
    g = load_golden("lss_small.npz")
    out64 = O.voxel_pooling(g["geom"], g["x"], g["dx"], g["bx"], g["nx"], exact=True)
    out32 = O.voxel_pooling(g["geom"], g["x"], g["dx"], g["bx"], g["nx"], exact=False)
    assert out64.shape == g["out"].shape
    # reference cumsum trick carries ~1e-5 error; north_star tolerance 1e-4
    assert np.abs(out64 - g["out"]).max() < 1e-4
    assert np.abs(out32 - g["out"]).max() < 1e-4
    assert np.abs(out64 - g["out_accelerated"]).max() < 1e-5
    # empty voxels stay exactly zero in both
    assert np.array_equal(out64 == 0, g["out_accelerated"] == 0)


def test_voxel_pooling_grad_small_vs_reference():
    g = load_golden("lss_small.npz")
    gx = O.voxel_pooling_grad(g["geom"], g["grad_out"], g["dx"], g["bx"], g["nx"], g["x"].shape[-1])
    assert np.array_equal(gx, g["grad_x"])  # pure gather: bit-exact


def test_lift_splat_small_vs_reference_module_forward():
    g = load_golden("lss_lift_small.npz")
    bev = O.lift_splat(g["depth"], g["img_feat"], g["geom"], g["dx"], g["bx"], g["nx"], exact=True)
    assert np.abs(bev - g["bev"]).max() < 1e-4


def test_bev_pool_contract_equals_voxel_pooling():
    """bev_pool(...) (ops/bev_pool) == voxel_pooling after the documented
    permute/transpose (view_transformer.py:166-169,186)."""
    g = load_golden("lss_small.npz")
    B, N, D, H, W, C = g["x"].shape
    idx, kept = O.voxel_index(g["geom"].reshape(-1, 3), g["dx"], g["bx"], g["nx"])
    b_ix = np.repeat(np.arange(B), idx.shape[0] // B)
    coords = np.concatenate([idx, b_ix[:, None]], 1)[kept]
    feats = g["x"].reshape(-1, C)[kept]
    nx = g["nx"].astype(int)
    out = O.bev_pool(feats, coords, B, nx[2], nx[0], nx[1], exact=True)   # [B,C,Z,X,Y]
    final = out.transpose(0, 1, 2, 4, 3)                                   # [B,C,Z,Y,X]
    final = np.concatenate([final[:, :, z] for z in range(nx[2])], 1)
    assert np.abs(final - g["out"]).max() < 1e-4
    # backward of the extension contract == gather
    f, gg, st, ln = O.bev_pool_prepare(feats, coords, B, nx[2], nx[0], nx[1])
    og = np.random.default_rng(0).normal(size=(B, nx[2], nx[0], nx[1], C)).astype(np.float32)
    xg = O.bev_pool_backward(og, gg, st, ln, f.shape[0])
    assert np.array_equal(xg, og[gg[:, 3], gg[:, 2], gg[:, 0], gg[:, 1]])
