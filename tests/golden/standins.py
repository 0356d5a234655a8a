"""The ONE plain-torch stand-in shared by tests/golden/make_golden.py (registered on the REFERENCE's stub registry) and the GPU
test (registered on the product's registry) for the detector-level BEVDepth4DDistill fixture: the image backbone, mmdet's
ResNet-50, which lives in the un-vendored mmdet package.  Everything else of the recipe is the reference's own code on one
side and the product's on the other: image neck (necks/fpn.py FPNForBEVDet), view transformer with its SE + BasicBlock depth
net + DCN, lift-splat, pre-process net / BEV encoder (backbones/resnet.py ResNetForBEVDet on bricks/res_block.py) and BEV neck
(necks/lss_fpn.py FPN_LSS), shift, depth loss, CenterHead, the whole teacher, the distillation losses.  No reference dependency."""
import torch.nn as nn
import torch.nn.functional as F


class TinyImageBackbone(nn.Module):
    """3 -> (C4, C5) channels at strides 16 / 32 (two convs, no norm): the two levels `out_indices=(2, 3)` hands the image neck"""

    def __init__(self, out_channels=(32, 64), **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(3, out_channels[0], 16, stride=16)
        self.conv2 = nn.Conv2d(out_channels[0], out_channels[1], 2, stride=2)

    def forward(self, x):
        c4 = F.relu(self.conv(x))
        return c4, F.relu(self.conv2(c4))


STANDINS = dict(TinyImageBackbone=TinyImageBackbone)


# ---- the small BEVDepth4DDistill recipe both sides build (same structure as configs/distillbev_centerpoint2bevdepth4d_r50.py) ----
PCR = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
         dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
         dict(num_class=2, class_names=["motorcycle", "bicycle"]), dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
COMMON_HEADS = dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))
INPUT_SIZE = (64, 176)                       # 4 x 11 feature map at stride 16
GRID = dict(xbound=[-51.2, 51.2, 3.2], ybound=[-51.2, 51.2, 3.2], zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 60.0, 1.0])    # 32 x 32 BEV
T_VOXEL = [0.8, 0.8, 8]                      # teacher pillars: 128 x 128 canvas -> SECOND strides 2, 4, 8 -> 64 / 32 / 16 -> neck 32 x 32


def head_cfg(in_channels, voxel_xy, out_size_factor):
    return dict(type="CenterHead", in_channels=in_channels, tasks=TASKS, common_heads=COMMON_HEADS, share_conv_channel=16,
                bbox_coder=dict(type="CenterPointBBoxCoder", pc_range=PCR[:2], post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                                max_num=500, score_threshold=0.1, out_size_factor=out_size_factor, voxel_size=voxel_xy, code_size=9),
                separate_head=dict(type="SeparateHead", init_bias=-2.19, final_kernel=3, head_conv=16),
                loss_cls=dict(type="GaussianFocalLoss", reduction="mean"),
                loss_bbox=dict(type="L1Loss", reduction="mean", loss_weight=0.25), norm_bbox=True)


def train_cfg(grid, voxel, out_size_factor):
    return dict(pts=dict(grid_size=[grid, grid, 1], voxel_size=voxel, out_size_factor=out_size_factor, dense_reg=1, gaussian_overlap=0.1,
                         max_objs=500, min_radius=2, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
                         point_cloud_range=PCR))


def teacher_cfg():
    return dict(model=dict(
        type="DynamicCenterPoint",
        pts_voxel_layer=dict(max_num_points=-1, voxel_size=T_VOXEL, max_voxels=(-1, -1), point_cloud_range=PCR),
        pts_voxel_encoder=dict(type="DynamicPillarFeatureNet", in_channels=5, feat_channels=[16], with_distance=False,
                               voxel_size=T_VOXEL, norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), point_cloud_range=PCR),
        pts_middle_encoder=dict(type="PointPillarsScatter", in_channels=16, output_shape=(128, 128)),
        pts_backbone=dict(type="SECOND", in_channels=16, out_channels=[16, 32, 64], layer_nums=[1, 1, 1], layer_strides=[2, 2, 2],
                          norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False)),
        pts_neck=dict(type="SECONDFPN", in_channels=[16, 32, 64], out_channels=[16, 16, 16], upsample_strides=[0.5, 1, 2],
                      norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                      use_conv_for_no_stride=True),
        pts_bbox_head=head_cfg(48, T_VOXEL[:2], 4),
        train_cfg=train_cfg(128, T_VOXEL, 4), test_cfg=None))


def distill_cfg(teacher, wide=False):
    """student: 2 frames x 6 cameras of 64 x 176 -> 32 x 32 BEV (24 channels per frame) -> BEV encoder -> 32 channels at 32 x 32.
    wide=True (round 5, bevdepth_step_wide.npz): the same step with a BEV encoder of the reference's OWN Bottleneck blocks
    (bricks/res_block.py:102-230 through backbones/resnet.py block_type='BottleNeck') at 64 / 256 channels and a 512 -> 64 BEV neck,
    i.e. channel counts the product's hand-written dense kernels take (1x1 layers: multiples of 64 -> bf16x6 GEMMs with statistics
    epilogues, 3x3 stride-1 layers with 64 outputs -> Winograd): the detector-level reference for those kernels."""
    cfg = _distill_cfg(teacher)
    if wide:
        cfg["pre_process"] = dict(type="ResNetForBEVDet", numC_input=32, num_layer=[1], num_channels=[32], stride=[1], backbone_output_ids=[0])
        cfg["img_view_transformer"]["numC_Trans"] = 32
        cfg["img_bev_encoder_backbone"] = dict(type="ResNetForBEVDet", numC_input=64, num_layer=[2, 1], num_channels=[256, 64],
                                               stride=[2, 2], block_type="BottleNeck")      # (a narrow second stage keeps the fixture small)
        cfg["img_bev_encoder_neck"] = dict(type="FPN_LSS", in_channels=256 + 64, out_channels=32, scale_factor=2, input_feature_index=(0, 1))
        cfg["distill_params"]["student_channels"] = [256, 32]
        cfg["distill_params"]["adaptation_type"] = ["1x1conv", "1x1conv"]
    return cfg


def _distill_cfg(teacher):
    return dict(
        type="BEVDepth4DDistill", teacher_config=teacher, teacher_ckpt=None, self_ckpt=None, inherit_head=False, distill_type="fgd",
        aligned=True, detach=True, before=True, interpolation_mode="bilinear",
        pre_process=dict(type="ResNetForBEVDet", numC_input=24, num_layer=[1], num_channels=[24], stride=[1], backbone_output_ids=[0]),
        distill_params=dict(
            student_channels=[16, 32], teacher_channels=[64, 48], spatial_t=0.5, spatial_student_ratio=1.0, channel_t=0.5,
            fg_feat_loss_weights=[3e-3, 2e-3], bg_feat_loss_weights=[4e-2, 3e-2], channel_loss_weights=[0.25], spatial_loss_weights=[1e-3, 1e-3],
            adaptation_type=["3x3conv", "1x1conv"], student_adaptation_params=dict(kernel_size=1, stride=1),
            teacher_adaptation_type="identity", teacher_adaptation_params=dict(kernel_size=4, stride=4),
            spatial_attentions=["teacher_student"], feat_criterion=dict(type="MSELoss", reduction="none"),
            spatial_criterion=dict(type="L1Loss", reduction="none"), channel_criterion=dict(type="L1Loss", reduction="none"),
            transpose_mask=False, foreground_mask="gt", background_mask="logical_not", scale_mask="combine_gt", spatial_mask=True,
            channel_mask=False, student_feat_pos=["backbone0", "head"], teacher_feat_pos=["backbone2", "head"], two_stage_epoch=-1,
            affinity_weights=[0, 0], affinity_mode="none", affinity_criterion=dict(type="SmoothL1Loss"), affinity_split=1,
            non_empty_weight=0, output_threshold=0.1, groundtruth_threshold=None, fp_as_foreground=["none", "teacher"], fp_weight=6e-2,
            fp_epoch=0, multi_scale_epoch=-1, fp_scale_mode="average", gauss_fg_weight=-1e10, context_length=0, context_weight=0),
        img_backbone=dict(type="TinyImageBackbone", out_channels=(32, 64)),
        img_neck=dict(type="FPNForBEVDet", in_channels=[32, 64], out_channels=32, num_outs=1, start_level=0, out_ids=[0]),
        img_view_transformer=dict(type="ViewTransformerLSSBEVDepth", loss_depth_weight=100.0, grid_config=GRID,
                                  data_config=dict(input_size=INPUT_SIZE), numC_input=32, numC_Trans=24,
                                  extra_depth_net=dict(type="ResNetForBEVDet", numC_input=16, num_layer=[2], num_channels=[16], stride=[1]),
                                  dcn_config=dict(bias=True)),
        img_bev_encoder_backbone=dict(type="ResNetForBEVDet", numC_input=48, num_channels=[16, 32, 64]),
        img_bev_encoder_neck=dict(type="FPN_LSS", in_channels=16 + 64, out_channels=32),
        pts_bbox_head=head_cfg(32, [0.1, 0.1], 32), train_cfg=train_cfg(1024, [0.1, 0.1, 0.2], 32), test_cfg=None)
