"""GPU: the BEVFormer distillation detector (BASELINE configs[4], distill_bev_amd/bevformer.py).

* FGD terms at the BEVFormer geometry (cell-centre coordinates, out_size_factor 512 / 20 = 25.6) against a fixture computed
  by the reference's own bevformer_distill.py (foreground_scale_mask :404-496, fgd_distill_loss :634-812): masks bit-equal
  (fg_scale 1 ulp: see ForegroundMaskRasterizer), losses and gradients 1e-4 -- on the unfused kernels AND through the fused
  adaptation + masked-MSE kernel when it is eligible;
* the whole detector at a reduced size (ResNet-18, width 32, 20 x 20 BEV, 3-frame queue, MVP virtual-point teacher on a
  41 x 160 x 160 grid): loss keys of the reference, finite, every trainable parameter reached, two seeded runs
  agree to 1e-4 (MIOpen's convolutions are not bit-reproducible), a Trainer step updates the student and leaves the hidden teacher untouched.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLD)
PCR = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]

DP = dict(student_channels=[32], teacher_channels=[32], spatial_t=0.5, spatial_student_ratio=1.0, channel_t=0.5,
          fg_feat_loss_weights=[3e-3], bg_feat_loss_weights=[4e-2], spatial_loss_weights=[1e-3], adaptation_type="1x1conv",
          student_adaptation_params=dict(kernel_size=1, stride=1), teacher_adaptation_type="identity",
          teacher_adaptation_params=dict(kernel_size=4, stride=4), spatial_attentions=["teacher"],
          feat_criterion=dict(type="MSELoss", reduction="none"), spatial_criterion=dict(type="L1Loss", reduction="none"),
          channel_criterion=dict(type="L1Loss", reduction="none"), transpose_mask=False, foreground_mask="gt",
          background_mask="logical_not", scale_mask="combine_gt", spatial_mask=True, channel_mask=False,
          student_feat_pos=["head"], teacher_feat_pos=["head"], affinity_mode="none", non_empty_weight=0, output_threshold=0.1,
          groundtruth_threshold=None, fp_as_foreground="none", fp_weight=0, fp_epoch=0, multi_scale_epoch=-1,
          fp_scale_mode="average", context_length=0, context_weight=0)


@pytest.mark.parametrize("fused", [False, True])
def test_bevformer_fgd_terms_vs_reference_fixture(fused):
    import copy
    from distill_bev_amd.bevformer import BEVFormerDistill
    from distill_bev_amd.center_head import LiDARBoxes
    from distill_bev_amd.detectors import install_fgd_modules
    from distill_bev_amd.distill_loss import ForegroundMaskRasterizer
    fx = np.load(os.path.join(GOLD, "bevformer_fgd.npz"))
    dev = torch.device("cuda:0")
    d = BEVFormerDistill.__new__(BEVFormerDistill)
    torch.nn.Module.__init__(d)
    d.teacher_model = torch.nn.Identity()
    d.distill_params = copy.deepcopy(DP)
    install_fgd_modules(d, d.distill_params)
    d._fg_raster = ForegroundMaskRasterizer([512, 512, 1], PCR, [0.2, 0.2, 8], cell_center=True)
    d._epoch, d.no_bg = 0, False
    d.fused_adapt_mse = fused
    d.channel_wise_adaptations.load_state_dict({k[5:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("cwa__")})
    d.spatial_wise_adaptations.load_state_dict({k[5:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("swa__")})
    d.to(dev)
    B = fx["teacher"].shape[0]
    gtb = [LiDARBoxes(fx[f"gt_boxes{b}"]) for b in range(B)]
    fg, fgs, bgs = d._fg_raster(20, 20, [b.tensor for b in gtb], dev)
    assert torch.equal(fg.cpu(), torch.from_numpy(fx["fg"]).float())
    assert torch.equal(bgs.cpu(), torch.from_numpy(fx["bg_scale"]))
    ref_s = torch.from_numpy(fx["fg_scale"])
    assert float((fgs.cpu() - ref_s).abs().max()) <= 2e-7 * float(ref_s.abs().max() + 1e-12)
    # channels-last views, as forward_distill hands the [bs, H*W, C] embeddings over
    t = torch.from_numpy(fx["teacher"]).to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    s = torch.from_numpy(fx["student"]).to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)
    losses = d.fgd_distill_loss(t, s, gtb, None, None, None, None, None, 0)
    assert set(losses) == {"kd_fg_feat_loss", "kd_bg_feat_loss", "kd_spatial_loss"}
    for k, v in losses.items():
        ref = float(fx["loss__" + k])
        assert abs(float(v) - ref) <= 1e-4 * abs(ref), (k, float(v), ref)
    params = list(d.parameters())
    grads = torch.autograd.grad(sum(losses.values()), [s] + params)
    refs = [fx["g_student"]] + [fx[f"g__{i}"] for i in range(len(params))]
    for g, r in zip(grads, refs):
        r = torch.from_numpy(r)
        assert float((g.cpu() - r).abs().max()) <= 1e-4 * max(float(r.abs().max()), 1e-8)


def test_bevformer_false_positive_mask_vs_reference_fixture():
    """add_fp_as_fg_bbox (bevformer_distill.py:555-631): cells inside teacher boxes scored above output_threshold and outside
    every ground-truth box, 1 / count scale, counts -- bit-equal to the reference's own function (including its [x, y]
    layout), and the FGD loss with fp_as_foreground='teacher' builds the extra term from it."""
    import copy
    from distill_bev_amd.bevformer import BEVFormerDistill
    from distill_bev_amd.center_head import LiDARBoxes
    from distill_bev_amd.detectors import install_fgd_modules
    from distill_bev_amd.distill_loss import ForegroundMaskRasterizer
    fx = np.load(os.path.join(GOLD, "bevformer_fgd.npz"))
    dev = torch.device("cuda:0")
    d = BEVFormerDistill.__new__(BEVFormerDistill)
    torch.nn.Module.__init__(d)
    d.teacher_model = torch.nn.Identity()
    d.distill_params = dict(copy.deepcopy(DP), fp_as_foreground="teacher", fp_weight=6e-2)
    install_fgd_modules(d, d.distill_params)
    d._fg_raster = ForegroundMaskRasterizer([512, 512, 1], PCR, [0.2, 0.2, 8], cell_center=True)
    d._epoch, d.no_bg, d.fused_adapt_mse = 0, False, False
    d.channel_wise_adaptations.load_state_dict({k[5:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("cwa__")})
    d.spatial_wise_adaptations.load_state_dict({k[5:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("swa__")})
    d.to(dev)
    B = 3
    gtb = [LiDARBoxes(fx[f"gt_boxes{b}"]) for b in range(B)]
    preds = [(LiDARBoxes(fx[f"t_boxes{b}"]), torch.from_numpy(fx[f"t_scores{b}"]).to(dev), None) for b in range(B)]
    fg, _, _ = d._fg_raster(20, 20, [b.tensor for b in gtb], dev)
    fp, fps, nfp = d.add_fp_as_fg("teacher", fg, None, preds, None)
    assert torch.equal(fp.cpu(), torch.from_numpy(fx["fp"]).float())
    assert torch.equal(nfp.cpu(), torch.from_numpy(fx["n_fp"]).float())
    assert float((fps.cpu() - torch.from_numpy(fx["fp_scale"]).float()).abs().max()) <= 1e-9
    t = torch.from_numpy(fx["teacher"]).to(dev)
    s = torch.from_numpy(fx["student"]).to(dev).requires_grad_(True)
    losses = d.fgd_distill_loss(t, s, gtb, None, None, None, preds, None, 0)
    assert set(losses) == {"kd_fg_feat_loss", "kd_bg_feat_loss", "kd_spatial_loss", "kd_fp_bg_feat_loss"}
    for k, v in losses.items():                     # the four terms of the reference's own fgd_distill_loss in this mode
        ref = float(fx["lossfp__" + k])
        assert abs(float(v) - ref) <= 1e-4 * abs(ref), (k, float(v), ref)


def small_distill_cfg(bev=20, dim=32, queue=3, cams=6):
    import bevformer_cfgs as C
    head = C.small_bevformer_head_cfg(dim=dim, bev=bev, levels=4, cams=cams, queries=40)
    train_cfg = dict(pts=head.pop("train_cfg"))
    head.pop("test_cfg")
    thead = C.small_dgcnn_head_cfg(dim=dim, bev=bev, levels=4, queries=40)
    thead["bbox_coder"]["voxel_size"] = [0.64, 0.64, 0.2]             # bev_shape = 102.4 / 0.64 = 160
    t_train = dict(pts=thead.pop("train_cfg"))
    thead.pop("test_cfg")
    teacher = dict(model=dict(
        type="MVPFormer",
        pts_voxel_encoder=dict(type="DynamicVoxelEncoder", pc_range=PCR, voxel_size=[0.64, 0.64, 0.2], virtual=True),
        pts_middle_encoder=dict(type="SparseEncoder", in_channels=23, sparse_shape=[41, 160, 160], output_channels=32,
                                order=("conv", "norm", "act"), encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 64), (64, 64)),
                                encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type="basicblock"),
        pts_backbone=dict(type="SECOND", in_channels=64, out_channels=[32, 64], layer_nums=[1, 1], layer_strides=[1, 2],
                          norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False)),
        pts_neck=dict(type="FPN", norm_cfg=dict(type="BN2d", eps=1e-3, momentum=0.01), act_cfg=dict(type="ReLU"),
                      in_channels=[32, 64], out_channels=dim, start_level=0, num_outs=4),
        pts_bbox_head=thead, train_cfg=t_train))
    return dict(
        type="BEVFormerDistill", inherit_head=True, inherit_decoder=True, inherit_query=False, teacher_config=teacher,
        teacher_ckpt=None, self_ckpt=None, distill_type="fgd", distill_params=dict(DP), use_grid_mask=True, video_test_mode=True,
        img_backbone=dict(type="ResNet", depth=18, num_stages=4, out_indices=(1, 2, 3), frozen_stages=-1,
                          norm_cfg=dict(type="BN", requires_grad=True), norm_eval=False, style="pytorch"),
        img_neck=dict(type="FPN", in_channels=[128, 256, 512], out_channels=dim, start_level=0, add_extra_convs="on_output",
                      num_outs=4, relu_before_extra_convs=True),
        pts_bbox_head=head, train_cfg=train_cfg)


def test_bevformer_distill_with_the_lidarformer_teacher_small():
    """The second shipped recipe (lidarformer_to_bevformer_nus_1x1conv_r50.py): hard voxelization (10 points / voxel) ->
    HardSimpleVFE -> SparseEncoder -> ... -> DGCNN3DHead teacher on plain 5-column LiDAR sweeps; one training forward + backward."""
    from distill_bev_amd import bevformer  # noqa: F401
    from distill_bev_amd import synthetic as syn
    from distill_bev_amd.registry import build_detector
    from distill_bev_amd.train_step import synthetic_teacher_checkpoint
    dev = torch.device("cuda:0")
    cfg = small_distill_cfg()
    t = cfg["teacher_config"]["model"]
    t["type"] = "LidarFormer"
    t["pts_voxel_layer"] = dict(max_num_points=10, voxel_size=[0.64, 0.64, 0.2], max_voxels=(9000, 12000), point_cloud_range=PCR)
    t["pts_voxel_encoder"] = dict(type="HardSimpleVFE", num_features=5)
    t["pts_middle_encoder"]["in_channels"] = 5
    cfg["teacher_ckpt"] = synthetic_teacher_checkpoint(cfg, seed=1)
    torch.manual_seed(1)
    model = build_detector(cfg)
    model.init_weights()
    model = model.to(dev).train()
    assert type(model.teacher_model).__name__ == "LidarFormer"
    batch = _batch(dev, seed=4)
    rng = np.random.default_rng(9)
    batch["points"] = [torch.from_numpy(syn.lidar_points(20000, rng)).to(dev) for _ in range(2)]
    torch.manual_seed(2)
    np.random.seed(2)
    losses = model.forward_train(**batch)
    assert set(losses) == {"loss_cls", "loss_bbox", "d0.loss_cls", "d0.loss_bbox", "kd_fg_feat_loss_head_head",
                           "kd_bg_feat_loss_head_head", "kd_spatial_loss_head_head"}
    assert all(bool(torch.isfinite(v)) for v in losses.values()), losses
    grads = torch.autograd.grad(sum(losses.values()), [p for p in model.parameters() if p.requires_grad], allow_unused=True)
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads)
    # the teacher's BEV embedding the FGD terms read is deterministic
    with torch.no_grad():
        e1 = model.teacher_model.pts_bbox_head(model.teacher_model.extract_feat(batch["points"], None, None)[1])["bev_embed"]
        e2 = model.teacher_model.pts_bbox_head(model.teacher_model.extract_feat(batch["points"], None, None)[1])["bev_embed"]
    assert e1.shape == (2, 400, 32) and float((e1 - e2).abs().max()) <= 1e-5 * float(e1.abs().max())
    os.remove(cfg["teacher_ckpt"])


def _build(seed=0):
    from distill_bev_amd import bevformer  # noqa: F401
    from distill_bev_amd.registry import build_detector
    from distill_bev_amd.train_step import synthetic_teacher_checkpoint
    cfg = small_distill_cfg()
    cfg["teacher_ckpt"] = synthetic_teacher_checkpoint(cfg, seed=seed)
    torch.manual_seed(seed)
    model = build_detector(cfg)
    model.init_weights()
    return model, cfg


def _batch(dev, seed=3):
    from distill_bev_amd.bevformer import make_bevformer_batch
    return make_bevformer_batch(2, np.random.default_rng(seed), dev, queue_length=3, n_cams=6, img_size=(96, 160),
                                n_points=(6000, 1500, 4000), n_boxes=8)


def test_bevformer_distill_step_small():
    from distill_bev_amd.config import Config
    from distill_bev_amd.train_step import Trainer
    dev = torch.device("cuda:0")
    model, cfg = _build()
    model = model.to(dev).train()
    assert not model.teacher_model.training and model.training
    t_before = {k: v.clone() for k, v in model.teacher_model.state_dict().items()}
    batch = _batch(dev)
    names = [n for n, p in model.named_parameters() if p.requires_grad]

    state = {k: v.clone() for k, v in model.state_dict().items()}

    def run():
        # the history frames run in eval mode on the BatchNorm running statistics the previous step updated: same state first
        model.load_state_dict(state)
        torch.manual_seed(11)
        np.random.seed(11)
        model.zero_grad(set_to_none=True)
        losses = model.forward_train(**batch)
        grads = torch.autograd.grad(sum(losses.values()), [p for p in model.parameters() if p.requires_grad], allow_unused=True)
        return losses, grads
    run()                       # first touch: MIOpen searches its solvers and may settle on different ones afterwards
    l1, g1 = run()
    assert set(l1) == {"loss_cls", "loss_bbox", "d0.loss_cls", "d0.loss_bbox", "kd_fg_feat_loss_head_head",
                       "kd_bg_feat_loss_head_head", "kd_spatial_loss_head_head"}
    assert all(bool(torch.isfinite(v)) for v in l1.values()), l1
    assert model.training and not model.teacher_model.training        # obtain_history_bev switched eval() -> train()
    unused = [n for n, g in zip(names, g1) if g is None]
    assert unused == [], unused
    assert all(bool(torch.isfinite(g).all()) for g in g1)
    l2, g2 = run()
    # MIOpen's convolutions (image backbone / neck, the two adaptation convs) pick split-K kernels with float atomics, so the
    # detector as a whole is stable rather than bit-identical run to run; the transformer / head / loss kernels of this library
    # ARE bit-reproducible on fixed inputs (test_gpu_bevformer_heads.py::test_bevformer_head_trains_and_is_reproducible)
    for k in l1:
        assert abs(float(l1[k]) - float(l2[k])) <= 1e-4 * abs(float(l1[k])), k
    for n, a, b in zip(names, g1, g2):
        assert float((a - b).abs().max()) <= 2e-3 * max(float(a.abs().max()), 1e-6), n
    # a full optimizer step through the Trainer (AdamW with the backbone at lr x 0.1, grad clip 35)
    tr = Trainer(model, Config(dict(optimizer=dict(type="AdamW", lr=2e-4, weight_decay=0.01,
                                                   paramwise_cfg=dict(custom_keys={"img_backbone": dict(lr_mult=0.1)})),
                                    optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)))), dev)
    before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    torch.manual_seed(11)
    np.random.seed(11)
    loss, _ = tr.step(batch)
    assert bool(torch.isfinite(loss))
    same = [n for n, p in model.named_parameters() if p.requires_grad and torch.equal(before[n], p.detach())]
    # zero_init_residual: the last norm of every residual block starts at weight 0, so on the first step nothing upstream of
    # it inside the block receives gradient -- the block's first norm bias (0, no decay effect) is the one tensor left as is
    assert all(n.startswith("img_backbone") and n.endswith("bn1.bias") for n in same), same
    for k, v in model.teacher_model.state_dict().items():
        assert torch.equal(v, t_before[k]), k
    # streaming inference on one sample: the BEV map of frame t feeds frame t + 1
    model.eval()
    meta = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in batch["img_metas"][0][2].items()}
    with torch.no_grad():
        r1 = model.forward_test([[meta]], [batch["img"][:1, 2]])
        assert model.prev_frame_info["prev_bev"] is not None and model.prev_frame_info["prev_bev"].shape == (1, 400, 32)
        r2 = model.forward_test([[dict(meta)]], [batch["img"][:1, 2]])
    for r in (r1, r2):
        assert set(r[0]["pts_bbox"]) == {"boxes_3d", "scores_3d", "labels_3d"}
        assert r[0]["pts_bbox"]["boxes_3d"].tensor.shape[1] == 9
    os.remove(cfg["teacher_ckpt"])


def test_bevformer_distill_forward_train_vs_reference_fixture():
    """The whole BEVFormerDistill.forward_train against the reference's own detector files (bevformer_distill.py on bevformer.py /
    mvx_two_stage.py / base.py; make_golden.py bevformer_step): GridMask draws, image branch (the shared two-conv stand-in for
    the un-vendored mmdet ResNet / FPN), history BEV over a 3-frame queue (frame 0 without history), BEVFormerHead + Hungarian
    losses, teacher DGCNN3DHead on the fixture's LiDAR pyramid, FGD terms.  The reference's state dict loads strict; the seven
    losses agree to 1e-4 and six spot-checked gradients (image branch, BEV queries, encoder, box branch, adaptation) to 1e-3."""
    import bevformer_cfgs as C
    from distill_bev_amd import bevformer  # noqa: F401
    from distill_bev_amd.center_head import LiDARBoxes
    from distill_bev_amd.registry import MODELS, build_detector
    for cls in (C.TinyBackbone, C.TinyNeck):
        MODELS.register_module(module=cls, force=True)
    fx = np.load(os.path.join(GOLD, "bevformer_step.npz"))
    dev = torch.device("cuda:0")
    head = C.no_dropout(C.small_bevformer_head_cfg(32, 10, 2, 3))
    thead = C.small_dgcnn_head_cfg(32, 10, 3)
    cfg = dict(type="BEVFormerDistill", inherit_head=False, inherit_decoder=False, inherit_query=False,
               teacher_config=dict(model=dict(type="MVPFormer", pts_bbox_head={k: v for k, v in thead.items() if k not in ("train_cfg", "test_cfg")},
                                              train_cfg=dict(pts=thead["train_cfg"]))),
               teacher_ckpt=None, distill_type="fgd", distill_params=dict(DP), use_grid_mask=True, video_test_mode=True,
               img_backbone=dict(type="TinyBackbone"), img_neck=dict(type="TinyNeck", out_channels=32),
               pts_bbox_head={k: v for k, v in head.items() if k not in ("train_cfg", "test_cfg")}, train_cfg=dict(pts=head["train_cfg"]))
    model = build_detector(cfg)
    sd = {k[7:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("model__")}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    model.teacher_model.pts_bbox_head.load_state_dict(
        {k[7:].replace("__", "."): torch.from_numpy(fx[k]) for k in fx.files if k.startswith("thead__")}, strict=True)
    model = model.to(dev).train()
    tfeats = [torch.from_numpy(fx[k]).to(dev) for k in ("tf0", "tf1", "tf2")]
    model.teacher_model.extract_feat = lambda points, img, img_metas: (None, tfeats)
    bs, queue, cams = 2, 3, 3
    H, W = [int(v) for v in fx["img_hw"]]
    metas = [{q: dict(can_bus=fx[f"can_bus_{b}_{q}"].copy(), lidar2img=list(fx[f"lidar2img_{b}_{q}"]), img_shape=[(H, W, 3)] * cams,
                      prev_bev_exists=q > 0, box_type_3d=lambda t, d=9: LiDARBoxes(t)) for q in range(queue)} for b in range(bs)]
    gtb = [LiDARBoxes(fx[f"gt_boxes{b}"]) for b in range(bs)]
    gtl = [torch.from_numpy(fx[f"gt_labels{b}"]).to(dev) for b in range(bs)]
    np.random.seed(7)
    torch.manual_seed(7)
    losses = model.forward_train(points=None, img_metas=metas, gt_bboxes_3d=gtb, gt_labels_3d=gtl, img=torch.from_numpy(fx["img"]).to(dev))
    assert list(losses) == ["loss_cls", "loss_bbox", "d0.loss_cls", "d0.loss_bbox", "kd_fg_feat_loss_head_head",
                            "kd_bg_feat_loss_head_head", "kd_spatial_loss_head_head"]
    got = {k: float(v) for k, v in losses.items()}
    want = {k: float(fx["loss__" + k.replace(".", "_")]) for k in losses}
    for k in losses:
        assert abs(got[k] - want[k]) <= 1e-4 * abs(want[k]), (got, want)
    names = [k[6:].replace("__", ".") for k in fx.files if k.startswith("grad__")]
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(sum(losses.values()), [params[n] for n in names])
    for n, g in zip(names, grads):
        ref = torch.from_numpy(fx["grad__" + n.replace(".", "__")])
        err = float((g.cpu() - ref).abs().max())
        assert err <= 1e-3 * float(ref.abs().max()), (n, err, float(ref.abs().max()))
    assert model.training and not model.teacher_model.pts_bbox_head.training
