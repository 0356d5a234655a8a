"""CPU: oracle/dynvoxel.py against the fixture the imported reference file computed (dynamic_voxel_encoder.py:8-102)."""
import os

import numpy as np

from oracle import dynvoxel as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dynvoxel.npz"))
CASES = ("mixed", "dense", "all_real", "all_virtual", "no_real", "no_virtual")


def test_plain_voxelization_matches_reference_bitwise():
    for i in range(2):
        v, c = O.voxelization(G[f"plain{i}_points"], G["pc_range"], G["voxel_size"])
        assert np.array_equal(c, G[f"plain{i}_coords"])
        assert np.array_equal(v, G[f"plain{i}_voxels"])


def test_virtual_voxelization_matches_reference_bitwise():
    for name in CASES:
        v, c = O.voxelization_virtual(G[f"virt_{name}_points"], G["pc_range"], G["voxel_size"])
        assert np.array_equal(c, G[f"virt_{name}_coords"]), name
        assert v.shape == G[f"virt_{name}_voxels"].shape
        assert np.array_equal(v, G[f"virt_{name}_voxels"]), (name, float(np.abs(v - G[f"virt_{name}_voxels"]).max()))


def test_fixture_covers_the_edge_cases():
    pts = G["virt_mixed_points"]
    assert (pts[0, :3] == G["pc_range"][3:]).all()            # a point on the closed upper border is kept ...
    assert G["virt_mixed_coords"].max(0).tolist() == [5, 16, 16]  # ... in cell index == shape
    v = G["virt_mixed_voxels"]
    mixed = (v[:, :6] != 0).any(1) & (v[:, 6:] != 0).any(1)
    assert mixed.sum() > 50
    assert (G["virt_all_real_voxels"][:, 6:] == 0).all() and (G["virt_all_virtual_voxels"][:, :6] == 0).all()
    assert np.bincount(np.unique(G["virt_dense_coords"], axis=0, return_inverse=True)[1].reshape(-1)).max() == 1


def test_encoder_batches_and_shape():
    v, c, shp = O.dynamic_voxel_encoder([G["virt_mixed_points"], G["virt_no_real_points"]], G["pc_range"], G["voxel_size"], True)
    assert np.array_equal(c, G["enc_coords"]) and np.array_equal(v, G["enc_voxels"]) and np.array_equal(shp, G["enc_shape"])
    v, c, shp = O.dynamic_voxel_encoder([G["plain0_points"], G["plain1_points"]], G["pc_range"], G["voxel_size"], False)
    assert np.array_equal(c, G["encp_coords"]) and np.array_equal(v, G["encp_voxels"]) and np.array_equal(shp, G["encp_shape"])
