"""GPU: sparse convolution kernels (csrc/spconv.hip) through the reference's module surface (SubMConv3d, SparseConv3d,
SparseInverseConv3d, SparseConv2d, SparseSequential, SparseEncoder, DynamicVoxelEncoder) against the fp64 oracle
(oracle/spconv.py: dense conv3d definition): output sites in the reference's order, features 2e-5 of the output scale,
gradients (features, weight) 1e-4; the reference's pair lists; run-to-run bit identity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(seed, B, shape, n, C):
    rng = np.random.default_rng(seed)
    vol = B * int(np.prod(shape))
    lin = rng.choice(vol, size=n, replace=False)
    rng.shuffle(lin)
    coords = np.stack(np.unravel_index(lin, (B, *shape)), 1).astype(np.int32)
    return coords, rng.normal(size=(n, C)).astype(np.float32)


@pytest.mark.parametrize("name,kw,Cin,Cout", [
    ("SubMConv3d", dict(kernel_size=3, padding=1), 16, 16),
    ("SubMConv3d", dict(kernel_size=3, padding=0), 23, 16),                  # conv_input of the MVP encoder: 23 channels, padding ignored
    ("SparseConv3d", dict(kernel_size=3, stride=2, padding=1), 16, 32),
    ("SparseConv3d", dict(kernel_size=3, stride=2, padding=[0, 1, 1]), 32, 64),
    ("SparseConv3d", dict(kernel_size=(3, 1, 1), stride=(2, 1, 1), padding=0), 64, 128),   # conv_out
    ("SparseConv3d", dict(kernel_size=3, stride=1, padding=2, dilation=2), 16, 24),
])
def test_sparse_conv3d_vs_dense_definition(name, kw, Cin, Cout):
    from distill_bev_amd import spconv
    from oracle import spconv as OS
    dev = torch.device("cuda:0")
    shape, B = (9, 14, 12), 2
    idx, feats = _cloud(11 + Cin, B, shape, 500, Cin)
    conv = getattr(spconv, name)(Cin, Cout, bias=True, indice_key="k1", **kw).to(dev)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev), list(shape), B)
    y = conv(x)
    subm = name.startswith("SubM")
    ref, oi, osh = OS.sparse_conv_dense(feats, idx, shape, B, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(),
                                        kw.get("stride", 1), kw.get("padding", 0), kw.get("dilation", 1), subm)
    assert list(y.spatial_shape) == list(osh)
    got_idx = y.indices.cpu().numpy().astype(np.int64)
    got = y.features.detach().cpu().double().numpy()
    if subm:
        order = np.lexsort((idx[:, 3], idx[:, 2], idx[:, 1], idx[:, 0]))
        assert np.array_equal(got_idx, idx.astype(np.int64))           # submanifold: rows stay in the input order
        got = got[order]
    else:
        assert np.array_equal(got_idx, oi.numpy())                    # strided: ascending cell order, as the reference
    assert np.abs(got - ref.numpy()).max() <= 2e-5 * np.abs(ref.numpy()).max()
    # dense() == the dense convolution itself
    dense = y.dense().cpu().double()
    assert dense.shape == (B, Cout, *osh)
    assert abs(float(dense.abs().sum()) - float(ref.abs().sum())) <= 1e-4 * float(ref.abs().sum())
    # the reference's pair lists are cached under the indice_key in its own tuple layout
    outids, in_idx, pairs, pair_num, sshape = x.indice_dict["k1"]
    _, opairs = OS.rulebook_pairs(idx, shape, B, conv.kernel_size, kw.get("stride", 1), kw.get("padding", 0), kw.get("dilation", 1), subm)
    nums = pair_num.cpu().tolist()
    assert nums == [len(p) for p in opairs] and pairs.shape == (int(np.prod(conv.kernel_size)), 2, 500)
    for k, n in enumerate(nums):
        a = set(map(tuple, pairs[k, :, :n].t().cpu().tolist()))
        assert a == set(map(tuple, opairs[k].tolist()))
        assert bool((pairs[k, :, n:] == -1).all())
    # gradients vs autograd through the dense fp64 definition
    g = torch.randn(y.features.shape, generator=torch.Generator().manual_seed(1))
    gf, gw, gb = torch.autograd.grad(y.features, (f, conv.weight, conv.bias), g.to(dev))
    f64 = torch.from_numpy(feats).double().requires_grad_(True)
    w64 = conv.weight.detach().cpu().double().requires_grad_(True)
    b64 = conv.bias.detach().cpu().double().requires_grad_(True)
    D, H, W = shape
    ii = torch.from_numpy(idx).long()
    dn = torch.zeros((B, D, H, W, Cin), dtype=torch.float64).index_put((ii[:, 0], ii[:, 1], ii[:, 2], ii[:, 3]), f64)
    dn = dn.permute(0, 4, 1, 2, 3)
    st, pd, dl = kw.get("stride", 1), kw.get("padding", 0), kw.get("dilation", 1)
    if subm:
        st, pd = 1, [k // 2 for k in conv.kernel_size]
    yy = torch.nn.functional.conv3d(dn, w64.permute(4, 3, 0, 1, 2), None, st, pd, dl)
    oi_t = torch.from_numpy(got_idx)
    rows = yy[oi_t[:, 0], :, oi_t[:, 1], oi_t[:, 2], oi_t[:, 3]] + b64
    rf, rw, rb = torch.autograd.grad(rows, (f64, w64, b64), g.double())
    for nm, a, b in (("features", gf, rf), ("weight", gw, rw), ("bias", gb, rb)):
        assert float((a.cpu().double() - b).abs().max()) <= 1e-4 * float(b.abs().max()), nm
    y2 = conv(spconv.SparseConvTensor(f.detach(), torch.from_numpy(idx).to(dev), list(shape), B))
    assert torch.equal(y2.features, y.features.detach())


def test_inverse_conv_and_2d_variants():
    from distill_bev_amd import spconv
    from oracle import spconv as OS
    dev = torch.device("cuda:0")
    shape, B = (8, 10, 9), 1
    idx, feats = _cloud(5, B, shape, 200, 16)
    down = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key="d").to(dev)
    up = spconv.SparseInverseConv3d(32, 16, 3, indice_key="d", bias=False).to(dev)
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), list(shape), B)
    mid = down(x)
    back = up(mid)
    assert torch.equal(back.indices, x.indices) and back.features.shape == (200, 16)
    # definition: out[i] = sum over the couple conv's pairs (i, o) of mid[o] @ W_inv[k]
    _, pairs = OS.rulebook_pairs(idx, shape, B, 3, 2, 1, 1, False)
    ref = np.zeros((200, 16))
    wi = up.weight.detach().cpu().double().numpy().reshape(27, 32, 16)
    m = mid.features.detach().cpu().double().numpy()
    for k, pr in enumerate(pairs):
        if len(pr):
            np.add.at(ref, pr[:, 0], m[pr[:, 1]] @ wi[k])
    assert np.abs(back.features.detach().cpu().double().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()
    # 2-D convolution = 3-D with a unit depth
    i2 = np.unique(idx[:, [0, 2, 3]], axis=0)
    f2 = np.random.default_rng(2).normal(size=(i2.shape[0], 16)).astype(np.float32)
    c2 = spconv.SparseConv2d(16, 16, 3, stride=2, padding=1, bias=False).to(dev)
    y2 = c2(spconv.SparseConvTensor(torch.from_numpy(f2).to(dev), torch.from_numpy(i2).to(dev), [10, 9], B))
    i3 = np.concatenate([i2[:, :1], np.zeros_like(i2[:, :1]), i2[:, 1:]], 1)
    w3 = c2.weight.detach().cpu().numpy()[None]
    ref2, oi2, osh2 = OS.sparse_conv_dense(f2, i3, (1, 10, 9), B, w3, None, (1, 2, 2), (0, 1, 1), 1, False)
    assert np.array_equal(y2.indices.cpu().numpy(), oi2.numpy()[:, [0, 2, 3]])
    assert np.abs(y2.features.detach().cpu().double().numpy() - ref2.numpy()).max() <= 2e-5 * np.abs(ref2.numpy()).max()


def test_sparse_encoder_of_the_mvp_teacher_vs_dense_reference_network():
    """SparseEncoder (configs/teacher_transformer/mvpformer.py:44-52 channel plan, small grid) in eval mode against the same
    network evaluated with dense fp64 convolutions and active-site masking (the definition of the sparse layers)."""
    from distill_bev_amd import sparse_encoder  # noqa: F401
    from distill_bev_amd.registry import build_middle_encoder
    from oracle import spconv as OS
    dev = torch.device("cuda:0")
    shape, B = [41, 24, 20], 2          # depth 41 -> 21 -> 11 -> 5 -> 2 (conv_out), as the full-size grid
    enc = build_middle_encoder(dict(
        type="SparseEncoder", in_channels=23, sparse_shape=shape, output_channels=128, order=("conv", "norm", "act"),
        encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
        encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type="basicblock")).to(dev).eval()
    g = torch.Generator().manual_seed(4)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    idx, feats = _cloud(9, B, shape, 1500, 23)
    with torch.no_grad():
        out = enc(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), B)
    # dense fp64 evaluation of the same layers
    def bn(m, x):
        return (x - m.running_mean.cpu().double()) / torch.sqrt(m.running_var.cpu().double() + m.eps) * m.weight.detach().cpu().double() \
            + m.bias.detach().cpu().double()

    def run(conv, f, ii, sh):
        subm = conv.subm
        o, oi, osh = OS.sparse_conv_dense(f, ii, sh, B, conv.weight.detach().cpu().numpy(), None, conv.stride, conv.padding,
                                          conv.dilation, subm)
        if subm:                                   # back to the input row order
            order = np.lexsort((ii[:, 3], ii[:, 2], ii[:, 1], ii[:, 0]))
            inv = np.empty_like(order); inv[order] = np.arange(len(order))
            return o[inv], ii, sh
        return o, oi.numpy(), osh

    f, ii, sh = torch.from_numpy(feats).double(), idx.astype(np.int64), shape
    f, ii, sh = run(enc.conv_input[0], f, ii, sh)
    f = torch.relu(bn(enc.conv_input[1], f))
    for stage in enc.encoder_layers:
        for blk in stage:
            if type(blk).__name__ == "SparseBasicBlock":
                idn = f
                o, _, _ = run(blk.conv1, f, ii, sh)
                o = torch.relu(bn(blk.norm1, o))
                o, _, _ = run(blk.conv2, o, ii, sh)
                f = torch.relu(bn(blk.norm2, o) + idn)
            else:
                f, ii, sh = run(blk[0], f, ii, sh)
                f = torch.relu(bn(blk[1], f))
    f, ii, sh = run(enc.conv_out[0], f, ii, sh)
    f = torch.relu(bn(enc.conv_out[1], f))
    dense = torch.zeros((B, 128, *sh), dtype=torch.float64)
    it = torch.from_numpy(np.asarray(ii)).long()
    dense[it[:, 0], :, it[:, 1], it[:, 2], it[:, 3]] = f
    ref = dense.view(B, 128 * sh[0], sh[1], sh[2])
    assert out.shape == ref.shape
    err = float((out.cpu().double() - ref).abs().max())
    assert err <= 1e-4 * float(ref.abs().max()), (err, float(ref.abs().max()))
    # the inference forward above ran conv + norm (+ identity) + relu as one kernel per layer; the autograd-recording forward
    # (separate BatchNorm1d / add / ReLU) gives the same features, and neither builds a pair list before a backward needs it
    from distill_bev_amd import spconv
    with torch.enable_grad():
        out2 = enc(torch.from_numpy(feats).to(dev).requires_grad_(True), torch.from_numpy(idx).to(dev), B)
    assert float((out2 - out).abs().max()) <= 2e-5 * float(ref.abs().max())
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), shape, B)
    with torch.no_grad():
        y = enc.encoder_layers(enc.conv_input(x))
    books = {id(rb): rb for rb in y.rulebooks.values()}
    assert len(books) == 7                    # 4 submanifold geometries (conv_input shares stage 1's) + 3 strided convolutions
    assert all(rb._pairs is None for rb in books.values())
    pairs = y.indice_dict["subm1"][2]          # on demand, in the reference's layout
    assert pairs.shape == (27, 2, 1500) and int(y.indice_dict["subm1"][3].sum()) == int((pairs[:, 0] >= 0).sum())
    sd = enc.state_dict()
    for k in ("conv_input.0.weight", "encoder_layers.encoder_layer1.0.conv1.weight", "encoder_layers.encoder_layer1.0.bn2.running_var",
              "encoder_layers.encoder_layer1.2.0.weight", "encoder_layers.encoder_layer4.1.conv2.weight", "conv_out.1.bias"):
        assert k in sd, k
    assert sd["conv_input.0.weight"].shape == (3, 3, 3, 23, 16)


def test_dynamic_voxel_encoder_vs_reference_fixture():
    """voxelization / voxelization_virtual / DynamicVoxelEncoder (dynamic_voxel_encoder.py:8-102) on the kernels
    (dbev_range_voxel_coords, dbev_virtual_voxel_reduce, dynamic-scatter grouping) against tests/golden/dynvoxel.npz, the
    outputs of the IMPORTED reference file: voxel coordinates and order exact, voxel rows BIT-equal (the kernels keep the
    reference's per-column summation order and its fp32 divisions), incl. points on the closed range border, clouds
    without real / without virtual points and voxels holding > 64 points of all three kinds."""
    import os
    from distill_bev_amd.sparse_encoder import DynamicVoxelEncoder, voxelization, voxelization_virtual
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dynvoxel.npz"))
    dev = torch.device("cuda:0")
    pr, vs = torch.from_numpy(G["pc_range"]), torch.from_numpy(G["voxel_size"])
    for i in range(2):
        v, c = voxelization(torch.from_numpy(G[f"plain{i}_points"]).to(dev), pr, vs)
        assert c.dtype == torch.int64 and np.array_equal(c.cpu().numpy(), G[f"plain{i}_coords"])
        assert np.array_equal(v.cpu().numpy(), G[f"plain{i}_voxels"])
    for name in ("mixed", "dense", "all_real", "all_virtual", "no_real", "no_virtual"):
        v, c = voxelization_virtual(torch.from_numpy(G[f"virt_{name}_points"]).to(dev), pr.to(dev), vs.to(dev))  # device triples too
        assert np.array_equal(c.cpu().numpy(), G[f"virt_{name}_coords"]), name
        ref = G[f"virt_{name}_voxels"]
        assert v.shape == ref.shape and np.array_equal(v.cpu().numpy(), ref), (name, float(np.abs(v.cpu().numpy() - ref).max()))
    v, c, shp = DynamicVoxelEncoder(G["pc_range"].tolist(), G["voxel_size"].tolist(), virtual=True)(
        [torch.from_numpy(G["virt_mixed_points"]).to(dev), torch.from_numpy(G["virt_no_real_points"]).to(dev)])
    assert np.array_equal(c.cpu().numpy(), G["enc_coords"]) and np.array_equal(v.cpu().numpy(), G["enc_voxels"])
    assert np.array_equal(shp, G["enc_shape"])
    v, c, shp = DynamicVoxelEncoder(G["pc_range"].tolist(), G["voxel_size"].tolist())(
        [torch.from_numpy(G["plain0_points"]).to(dev), torch.from_numpy(G["plain1_points"]).to(dev)])
    assert np.array_equal(c.cpu().numpy(), G["encp_coords"]) and np.array_equal(v.cpu().numpy(), G["encp_voxels"])
    # empty cloud / everything outside the range
    far = torch.full((10, 17), 99.0, device=dev)
    v, c = voxelization_virtual(far, pr, vs)
    assert v.shape == (0, 23) and c.shape == (0, 3)


def test_dynamic_voxel_encoder_full_size_vs_oracle():
    """400 k MVP points on the 1440 x 1440 x 40 grid of the shipped recipe (configs/teacher_transformer/mvpformer.py):
    coordinates exact and voxel rows bit-equal to the pinned CPU oracle (oracle/dynvoxel.py), bit-identical when repeated."""
    from distill_bev_amd.sparse_encoder import voxelization_virtual
    from oracle import dynvoxel as O
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    n = 400_000
    p = rng.normal(size=(n, 17)).astype(np.float32)
    p[:, :2] = rng.normal(0, 18, (n, 2)); p[:, 2] = rng.uniform(-5.5, 3.5, n)
    p[:, -2] = rng.choice(np.array([1, 0, -1, -1], np.float32), n)
    pcr, vs = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], [0.075, 0.075, 0.2]
    v, c = voxelization_virtual(torch.from_numpy(p).to(dev), pcr, vs)
    rv, rc = O.voxelization_virtual(p, pcr, vs)
    assert np.array_equal(c.cpu().numpy(), rc) and np.array_equal(v.cpu().numpy(), rv)
    v2, c2 = voxelization_virtual(torch.from_numpy(p).to(dev), pcr, vs)
    assert torch.equal(v, v2) and torch.equal(c, c2)


def test_edge_cases_empty_and_single_site_tensors():
    """No active site at all, and one active site: every sparse layer type returns the right (possibly empty) site set; the
    encoder of an empty sample is the zero canvas."""
    from distill_bev_amd import spconv
    dev = torch.device("cuda:0")
    shape, B = [9, 12, 10], 2
    for n in (0, 1):
        idx = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        if n:
            idx[0] = torch.tensor([1, 4, 5, 6], dtype=torch.int32)
        x = spconv.SparseConvTensor(torch.randn((n, 16), device=dev), idx, shape, B)
        with torch.no_grad():
            y = spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="a").to(dev)(x)
            z = spconv.SparseConv3d(32, 16, 3, stride=2, padding=1, bias=True, indice_key="b").to(dev)(y)
            w = spconv.SparseInverseConv3d(16, 8, 3, indice_key="b", bias=False).to(dev)(z)
        assert y.features.shape == (n, 32) and w.features.shape == (n, 8)
        # site (4, 5, 6) under a stride-2 / pad-1 / 3-tap convolution: outputs o with 2 o - 1 + k = coordinate -> z {2}, y {2, 3}, x {3}
        assert z.features.shape[0] == (0 if n == 0 else 2) and list(z.spatial_shape) == [5, 6, 5]
        if n:
            assert z.indices.cpu().tolist() == [[1, 2, 2, 3], [1, 2, 3, 3]]
        d = z.dense()
        assert d.shape == (B, 16, 5, 6, 5) and int((d != 0).any(dim=1).sum()) == z.features.shape[0]
        assert bool(torch.isfinite(d).all())


@pytest.mark.parametrize("n,Cin,Cout,kind", [(3000, 16, 16, "subm"), (3000, 32, 64, "down"), (2500, 64, 128, "down"), (1500, 128, 128, "subm"),
                                               (2000, 23, 16, "subm"), (1200, 48, 24, "down"), (800, 32, 16, "inverse"),
                                               (900, 192, 144, "subm"), (700, 256, 64, "down")])   # wider than the kernels' 128-channel tiles
def test_sparse_conv_backward_kernels_vs_pair_list_definition(n, Cin, Cout, kind):
    """indice_conv_backward (spconv_ops.h:352-420) on the kernels (dbev_spconv_backward_data / _weight): input and weight gradients
    against the fp64 definition summed over the oracle's pair lists (oracle/spconv.py rulebook_pairs), 1e-5 of scale; bit-identical
    when repeated (no float atomics); regular, submanifold and inverse layers, channel counts that need padding, and layers wider
    than 128 channels (run per 128-channel slice pair, ADVICE r3: they used to assert)."""
    from distill_bev_amd import spconv
    from oracle import spconv as OS
    dev = torch.device("cuda:0")
    shape, B = (10, 18, 16), 2
    idx, feats = _cloud(100 + n + Cin, B, shape, n, Cin if kind != "inverse" else 16)
    rng = np.random.default_rng(n)
    if kind == "inverse":
        down = spconv.SparseConv3d(16, Cin, 3, stride=2, padding=1, bias=False, indice_key="d").to(dev)
        conv = spconv.SparseInverseConv3d(Cin, Cout, 3, indice_key="d", bias=False).to(dev)
        x0 = spconv.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), list(shape), B)
        with torch.no_grad():
            mid = down(x0)
        f = mid.features.detach().clone().requires_grad_(True)
        x = spconv.SparseConvTensor(f, mid.indices, mid.spatial_shape, B)
        x.indice_dict, x.rulebooks = mid.indice_dict, mid.rulebooks
        _, pairs = OS.rulebook_pairs(idx, shape, B, 3, 2, 1, 1, False)
        pairs = [pr[:, ::-1] for pr in pairs]                       # (row of this layer's input, row of its output)
    else:
        subm = kind == "subm"
        kw = dict(kernel_size=3, padding=1) if subm else dict(kernel_size=3, stride=2, padding=1)
        conv = getattr(spconv, "SubMConv3d" if subm else "SparseConv3d")(Cin, Cout, bias=False, **kw).to(dev)
        f = torch.from_numpy(feats).to(dev).requires_grad_(True)
        x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev), list(shape), B)
        _, pairs = OS.rulebook_pairs(idx, shape, B, 3, 1 if subm else 2, 1, 1, subm)
    runs = []
    for _ in range(2):
        y = conv(x)
        g = torch.from_numpy(np.random.default_rng(7).normal(size=tuple(y.features.shape)).astype(np.float32)).to(dev)
        gf, gw = torch.autograd.grad(y.features, (f, conv.weight), g)
        runs.append((gf, gw))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    f64, g64 = f.detach().cpu().double().numpy(), g.cpu().double().numpy()
    w64 = conv.weight.detach().cpu().double().numpy().reshape(27, Cin, Cout)
    rf, rw = np.zeros_like(f64), np.zeros_like(w64)
    for k, pr in enumerate(pairs):
        if len(pr):
            rw[k] = f64[pr[:, 0]].T @ g64[pr[:, 1]]
            np.add.at(rf, pr[:, 0], g64[pr[:, 1]] @ w64[k].T)
    gf, gw = runs[0][0].cpu().double().numpy(), runs[0][1].cpu().double().numpy().reshape(27, Cin, Cout)
    assert np.abs(gf - rf).max() <= 1e-5 * np.abs(rf).max(), np.abs(gf - rf).max() / np.abs(rf).max()
    assert np.abs(gw - rw).max() <= 1e-5 * np.abs(rw).max(), np.abs(gw - rw).max() / np.abs(rw).max()


def test_indice_conv_api_backward_and_empty_inputs():
    """ops.indice_conv with the reference's arguments (pair lists) is differentiable through the same kernels; a layer without any
    active site returns empty / zero gradients."""
    from distill_bev_amd import spconv
    dev = torch.device("cuda:0")
    shape, B = (8, 12, 10), 1
    idx, feats = _cloud(3, B, shape, 400, 32)
    it = torch.from_numpy(idx).to(dev)
    outids, pairs, nums = spconv.get_indice_pairs(it, B, list(shape), 3, 2, 1, 1)
    w = torch.randn(3, 3, 3, 32, 64, device=dev, requires_grad=True)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    y = spconv.indice_conv(f, w, pairs, nums, outids.shape[0])
    conv = spconv.SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    f2 = f.detach().clone().requires_grad_(True)
    y2 = conv(spconv.SparseConvTensor(f2, it, list(shape), B)).features
    assert torch.equal(y, y2)
    g = torch.randn_like(y)
    a = torch.autograd.grad(y, (f, w), g)
    b = torch.autograd.grad(y2, (f2, conv.weight), g)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    e = spconv.SparseConvTensor(torch.zeros((0, 32), device=dev, requires_grad=True), torch.zeros((0, 4), dtype=torch.int32, device=dev), list(shape), B)
    ye = conv(e).features
    assert ye.shape == (0, 64)
    ge = torch.autograd.grad(ye.sum(), conv.weight, allow_unused=True)[0]
    assert ge is None or float(ge.abs().max()) == 0.0


@pytest.mark.parametrize("ndim,kw,Cin,Cout", [
    (3, dict(kernel_size=3, stride=2, padding=1), 32, 64),
    (3, dict(kernel_size=2, stride=2, padding=0), 16, 32),            # the usual up-sampling layer: every input owns a 2x2x2 block
    (3, dict(kernel_size=(3, 1, 3), stride=(1, 2, 2), padding=(1, 0, 1)), 16, 16),
    (3, dict(kernel_size=3, stride=1, padding=0, dilation=2), 16, 48),
    (2, dict(kernel_size=3, stride=2, padding=1), 16, 32),
])
def test_sparse_conv_transpose_vs_dense_definition_and_pair_list_gradients(ndim, kw, Cin, Cout):
    """SparseConvTranspose2d / 3d (conv.py:300-346; rulebook indice.h:88-140: out = in * stride - padding + k * dilation): output
    sites (ascending cell order), features against the dense fp64 conv_transpose3d of the densified input (oracle/spconv.py
    sparse_deconv_dense) at 1e-5 of scale, input / weight gradients against the fp64 sums over the oracle's pair lists, repeated runs
    bit-identical."""
    from distill_bev_amd import spconv
    from oracle import spconv as OS
    dev = torch.device("cuda:0")
    shape3, B = (6, 9, 8), 2
    idx, feats = _cloud(31 + Cin, B, shape3, 260, Cin)
    if ndim == 2:
        idx = idx[idx[:, 1] == 0]
        feats = feats[: idx.shape[0]]
        idx_in, shape_in = idx[:, [0, 2, 3]], list(shape3[1:])
        idx3 = idx.copy()
        shape_or = (1, *shape3[1:])
    else:
        idx_in, shape_in, idx3, shape_or = idx, list(shape3), idx, shape3
    cls = spconv.SparseConvTranspose3d if ndim == 3 else spconv.SparseConvTranspose2d
    conv = cls(Cin, Cout, bias=True, **kw).to(dev)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    x = spconv.SparseConvTensor(f, torch.from_numpy(np.ascontiguousarray(idx_in)).to(dev), shape_in, B)
    t3 = lambda v, fill: [int(q) for q in v] if isinstance(v, (list, tuple)) else [int(v)] * 3
    ks, st, pd, dl = kw["kernel_size"], kw.get("stride", 1), kw.get("padding", 0), kw.get("dilation", 1)
    if ndim == 2:
        ks, st, pd, dl = (1, ks, ks), (1, st, st), (0, pd, pd), (1, dl, dl)
    w = conv.weight.detach().cpu().double().numpy()
    w5 = w if ndim == 3 else w[None]
    ref, oi, osh = OS.sparse_deconv_dense(feats, idx3, shape_or, B, w5, conv.bias.detach().cpu().double().numpy(), st, pd, dl, 0)
    runs = []
    for _ in range(2):
        y = conv(x)
        g = torch.from_numpy(np.random.default_rng(3).normal(size=tuple(y.features.shape)).astype(np.float32)).to(dev)
        runs.append((y.features.detach().clone(),) + torch.autograd.grad(y.features, (f, conv.weight), g))
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1]))
    assert list(y.spatial_shape) == list(osh[3 - ndim:])
    got_idx = y.indices.cpu().numpy().astype(np.int64)
    want_idx = oi.numpy() if ndim == 3 else oi.numpy()[:, [0, 2, 3]]
    assert np.array_equal(got_idx, want_idx) and len(want_idx) > len(idx3)
    refn = ref.numpy()
    assert np.abs(runs[0][0].cpu().double().numpy() - refn).max() <= 1e-5 * np.abs(refn).max()
    _, pairs = OS.rulebook_pairs(idx3, shape_or, B, t3(ks, 1), st, pd, dl, False, True, 0)
    K = len(pairs)
    f64, g64, w64 = feats.astype(np.float64), g.cpu().double().numpy(), w.reshape(K, Cin, Cout)
    rf, rw = np.zeros_like(f64), np.zeros_like(w64)
    for k, pr in enumerate(pairs):
        if len(pr):
            rw[k] = f64[pr[:, 0]].T @ g64[pr[:, 1]]
            np.add.at(rf, pr[:, 0], g64[pr[:, 1]] @ w64[k].T)
    gf, gw = runs[0][1].cpu().double().numpy(), runs[0][2].cpu().double().numpy().reshape(K, Cin, Cout)
    assert np.abs(gf - rf).max() <= 1e-5 * np.abs(rf).max()
    assert np.abs(gw - rw).max() <= 1e-5 * np.abs(rw).max()


@pytest.mark.parametrize("ndim,C,kw", [(3, 32, dict(kernel_size=3, stride=2, padding=1)), (3, 6, dict(kernel_size=2, stride=2)),
                                       (2, 16, dict(kernel_size=3, stride=2, padding=1)), (3, 64, dict(kernel_size=3, stride=1, padding=1))])
def test_sparse_max_pool_forward_and_tie_gradients_bit_exact(ndim, C, kw):
    """SparseMaxPool2d / 3d (pool.py:21-88, pool_ops.h:26-98): the reference raises a ZERO-initialised output with every paired input
    (so all-negative windows give 0) and hands an output's gradient to EVERY input equal to its value; features quantised to halves
    so that ties happen.  Output sites, values and input gradients are bit-exact against oracle/spconv.py's pair-list restatement
    (sums of at most K gradients in offset order on both sides); the functional surface (indice_maxpool on the reference's pair
    lists) gives the same tensors."""
    from distill_bev_amd import spconv
    from oracle import spconv as OS
    dev = torch.device("cuda:0")
    shape3, B = (7, 10, 9), 2
    idx, feats = _cloud(57 + C, B, shape3, 400, C)
    feats = (np.round(feats * 2) / 2).astype(np.float32)
    if ndim == 2:
        idx = idx[idx[:, 1] == 0]
        feats = feats[: idx.shape[0]]
        idx_in, shape_in, shape_or = idx[:, [0, 2, 3]], list(shape3[1:]), (1, *shape3[1:])
        ks, st, pd = (1, kw["kernel_size"], kw["kernel_size"]), (1, kw.get("stride", 1), kw.get("stride", 1)), (0, kw.get("padding", 0), kw.get("padding", 0))
    else:
        idx_in, shape_in, shape_or = idx, list(shape3), shape3
        ks, st, pd = kw["kernel_size"], kw.get("stride", 1), kw.get("padding", 0)
    pool = (spconv.SparseMaxPool3d if ndim == 3 else spconv.SparseMaxPool2d)(**kw)
    f = torch.from_numpy(feats).to(dev).requires_grad_(True)
    x = spconv.SparseConvTensor(f, torch.from_numpy(np.ascontiguousarray(idx_in)).to(dev), shape_in, B)
    y = pool(x)
    out_idx, pairs = OS.rulebook_pairs(idx, shape_or, B, ks, st, pd, 1, False)
    want_idx = out_idx if ndim == 3 else out_idx[:, [0, 2, 3]]
    assert np.array_equal(y.indices.cpu().numpy().astype(np.int64), want_idx)
    ref = OS.maxpool_from_pairs(feats, pairs, len(out_idx))
    assert np.array_equal(y.features.detach().cpu().numpy().astype(np.float64), ref) and (ref == 0).any()
    g = (np.round(np.random.default_rng(2).normal(size=ref.shape) * 4) / 4).astype(np.float32)      # exactly summable
    (gf,) = torch.autograd.grad(y.features, f, torch.from_numpy(g).to(dev))
    rf = OS.maxpool_backward_from_pairs(feats, ref, g, pairs)
    assert np.array_equal(gf.cpu().numpy().astype(np.float64), rf) and np.abs(rf).max() > 0
    # the reference's functional entry on its own pair lists
    f2 = torch.from_numpy(feats).to(dev).requires_grad_(True)
    rb = spconv.build_rulebook(x.indices, B, shape_in, kw["kernel_size"], kw.get("stride", 1), kw.get("padding", 0), 1, False)
    y2 = spconv.indice_maxpool(f2, rb.indice_pairs, rb.indice_pair_num, rb.n_out)
    (gf2,) = torch.autograd.grad(y2, f2, torch.from_numpy(g).to(dev))
    assert torch.equal(y2, y.features) and torch.equal(gf2, gf)
