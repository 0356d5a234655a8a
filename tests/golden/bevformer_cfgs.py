"""Small instances of the head configs of the reference's BEVFormer distillation recipes
(configs/lidar2camera_bev_distillation/teacher_to_bevformer/*.py:84-165, configs/teacher_transformer/mvpformer.py:68-135): the
same dict structure and type names at width 32 / a 10 x 10 BEV grid.  Shared by make_golden.py (which builds them with the
REFERENCE's classes) and the GPU tests (which build them with the product's) -- data only, no reference dependency."""
PCR = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]


def _ffn(dim):
    # the layer classes default to ffn_cfgs with embed_dims=256: a smaller width has to name it
    return dict(type="FFN", embed_dims=dim, feedforward_channels=2 * dim, num_fcs=2, ffn_drop=0.1, act_cfg=dict(type="ReLU", inplace=True))


def small_bevformer_head_cfg(dim=32, bev=10, levels=2, cams=3, queries=12, enc_layers=2, dec_layers=2):
    return dict(
        type="BEVFormerHead", bev_h=bev, bev_w=bev, num_query=queries, num_classes=10, in_channels=dim, sync_cls_avg_factor=True,
        with_box_refine=True, as_two_stage=False,
        transformer=dict(
            type="PerceptionTransformer", rotate_prev_bev=True, use_shift=True, use_can_bus=True, embed_dims=dim,
            num_feature_levels=levels, num_cams=cams, rotate_center=[bev // 2, bev // 2],
            encoder=dict(type="BEVFormerEncoder", num_layers=enc_layers, pc_range=PCR, num_points_in_pillar=4,
                         return_intermediate=False,
                         transformerlayers=dict(
                             type="BEVFormerLayer",
                             attn_cfgs=[dict(type="TemporalSelfAttention", embed_dims=dim, num_heads=4, num_levels=1),
                                        dict(type="SpatialCrossAttention", pc_range=PCR, num_cams=cams,
                                             deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=dim, num_heads=4,
                                                                       num_points=8, num_levels=levels),
                                             embed_dims=dim)],
                             feedforward_channels=2 * dim, ffn_dropout=0.1, ffn_cfgs=_ffn(dim),
                             operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm"))),
            decoder=dict(type="DetectionTransformerDecoder", num_layers=dec_layers, return_intermediate=True,
                         transformerlayers=dict(
                             type="DetrTransformerDecoderLayer",
                             attn_cfgs=[dict(type="MultiheadAttention", embed_dims=dim, num_heads=4, dropout=0.1),
                                        dict(type="CustomMSDeformableAttention", embed_dims=dim, num_heads=4, num_levels=1)],
                             feedforward_channels=2 * dim, ffn_dropout=0.1, ffn_cfgs=_ffn(dim),
                             operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")))),
        bbox_coder=dict(type="NMSFreeCoder", post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], pc_range=PCR, max_num=8,
                        voxel_size=[0.2, 0.2, 8], num_classes=10),
        positional_encoding=dict(type="LearnedPositionalEncoding", num_feats=dim // 2, row_num_embed=bev, col_num_embed=bev),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_bbox=dict(type="L1Loss", loss_weight=0.25), loss_iou=dict(type="GIoULoss", loss_weight=0.0),
        train_cfg=dict(grid_size=[512, 512, 1], voxel_size=[0.2, 0.2, 8], point_cloud_range=PCR, out_size_factor=4,
                       assigner=dict(type="HungarianAssigner3D", cls_cost=dict(type="FocalLossCost", weight=2.0),
                                     reg_cost=dict(type="BBox3DL1Cost", weight=0.25), iou_cost=dict(type="IoUCost", weight=0.0),
                                     pc_range=PCR)),
        test_cfg=None)


def small_dgcnn_head_cfg(dim=32, bev=10, levels=3, queries=12, enc_layers=2, dec_layers=2):
    return dict(
        type="DGCNN3DHead", num_query=queries, num_classes=10, in_channels=dim, sync_cls_avg_factor=True, with_box_refine=True,
        as_two_stage=False, bev_h=bev, bev_w=bev,
        transformer=dict(
            type="DeformableDetrTransformerDistill", bev_size=bev, num_feature_levels=levels,
            encoder=dict(type="DetrTransformerEncoder", num_layers=enc_layers,
                         transformerlayers=dict(type="BaseTransformerLayer",
                                                attn_cfgs=dict(type="MultiScaleDeformableAttention", embed_dims=dim, num_heads=4,
                                                               num_levels=levels),
                                                feedforward_channels=2 * dim, ffn_dropout=0.1, ffn_cfgs=_ffn(dim),
                                                operation_order=("cross_attn", "norm", "ffn", "norm"))),
            decoder=dict(type="DetectionTransformerDecoder", num_layers=dec_layers, return_intermediate=True,
                         transformerlayers=dict(
                             type="DetrTransformerDecoderLayer",
                             attn_cfgs=[dict(type="MultiheadAttention", embed_dims=dim, num_heads=4, dropout=0.1),
                                        dict(type="CustomMSDeformableAttention", embed_dims=dim, num_heads=4, num_levels=1)],
                             feedforward_channels=2 * dim, ffn_dropout=0.1, ffn_cfgs=_ffn(dim),
                             operation_order=("self_attn", "norm", "cross_attn", "norm", "ffn", "norm")))),
        bbox_coder=dict(type="NMSFreeCoder", post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], pc_range=PCR, max_num=8,
                        voxel_size=[3.2, 3.2, 0.2], num_classes=10),
        positional_encoding=dict(type="SinePositionalEncoding", num_feats=dim // 2, normalize=True, offset=-0.5),
        loss_cls=dict(type="FocalLoss", use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=2.0),
        loss_bbox=dict(type="L1Loss", loss_weight=0.25), loss_iou=dict(type="GIoULoss", loss_weight=0.0),
        train_cfg=dict(grid_size=[32, 32, 1], voxel_size=[3.2, 3.2, 0.2], point_cloud_range=PCR, out_size_factor=4,
                       assigner=dict(type="HungarianAssigner3D", cls_cost=dict(type="FocalLossCost", weight=2.0),
                                     reg_cost=dict(type="BBox3DL1Cost", weight=0.25), iou_cost=dict(type="IoUCost", weight=0.0),
                                     pc_range=PCR)),
        test_cfg=None)


# ---- detector-level fixture (make_golden.py bevformer_step): stand-ins for the un-vendored image branch -----------------
# mmdet's ResNet / FPN are not in the reference tree; the detector-level fixture only needs SOME image branch that both sides
# share, so these two plain-torch modules are registered on the reference's stub registries by make_golden.py and on the
# product's registry by the test.  (Data module: no reference dependency.)
import torch.nn as _nn


class TinyBackbone(_nn.Module):
    """3 -> 16 channels at stride 8 and stride 16 (no normalisation: nothing changes between train() and eval())"""

    def __init__(self, **kwargs):
        super().__init__()
        self.c1 = _nn.Conv2d(3, 16, 8, stride=8)
        self.c2 = _nn.Conv2d(16, 16, 2, stride=2)

    def forward(self, x):
        a = _nn.functional.relu(self.c1(x))
        return a, _nn.functional.relu(self.c2(a))


class TinyNeck(_nn.Module):
    def __init__(self, out_channels=32, **kwargs):
        super().__init__()
        self.l1 = _nn.Conv2d(16, out_channels, 1)
        self.l2 = _nn.Conv2d(16, out_channels, 1)

    def forward(self, feats):
        return [self.l1(feats[0]), self.l2(feats[1])]


_DROPOUT_BY_DEFAULT = ("TemporalSelfAttention", "SpatialCrossAttention", "CustomMSDeformableAttention", "MultiScaleDeformableAttention",
                       "MSDeformableAttention3D")


def no_dropout(cfg):
    """the same config with every dropout probability at 0 (train-mode fixtures must not depend on the RNG stream); the
    attention types whose constructors default to dropout 0.1 get the key spelled out"""
    if isinstance(cfg, dict):
        out = {k: (0.0 if k in ("dropout", "ffn_dropout", "ffn_drop", "attn_drop", "proj_drop") else no_dropout(v)) for k, v in cfg.items()}
        if out.get("type") in _DROPOUT_BY_DEFAULT:
            out["dropout"] = 0.0
        return out
    if isinstance(cfg, (list, tuple)):
        return type(cfg)(no_dropout(v) for v in cfg)
    return cfg
