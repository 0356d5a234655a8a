# DistillBEV: CenterPoint (pillar, dynamic voxelization) teacher -> BEVDepth4D-R50 student, FGD loss.
#
# The EFFECTIVE recipe of the reference, i.e. its
#   configs/lidar2camera_bev_distillation/centerpoint_pillar_to_bevdepth4d_r50/
#       centerpoint_02pillar_second_secfpn_circlenms_8x4_cyclic_20e_nus_to_bevdepth4d_r50.py
# with the --cfg-options of scripts/teacher_to_bevdepth4d/centerpoint2bevdepth.sh:23-47 applied and the
# teacher config (configs/dynamic_centerpoint/dynamic_centerpoint_02pillar_second_secfpn_4x8_cyclic_20e_nus.py
# + its _base_ chain) resolved inline -- model section only (data pipelines / schedules are outside the hot
# path; bench.py feeds synthetic nuScenes-shaped tensors).  tests/test_config.py checks this file against the
# reference's own config files loaded through distill_bev_amd.config where /root/reference is available.
#
# Differences that are deliberate MI355X choices, not model changes:
#   img_backbone.with_cp=False   activation checkpointing is a memory/recompute trade; 288 GB HBM3E holds the
#                                bs=8 x 12-image activations, so nothing is recomputed.
#   *_ckpt / pretrained = None   no checkpoints are reachable (no network); weights are seeded random init.

point_cloud_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
class_names = ['car', 'truck', 'construction_vehicle', 'bus', 'trailer', 'barrier',
               'motorcycle', 'bicycle', 'pedestrian', 'traffic_cone']
tasks = [dict(num_class=1, class_names=['car']),
         dict(num_class=2, class_names=['truck', 'construction_vehicle']),
         dict(num_class=2, class_names=['bus', 'trailer']),
         dict(num_class=1, class_names=['barrier']),
         dict(num_class=2, class_names=['motorcycle', 'bicycle']),
         dict(num_class=2, class_names=['pedestrian', 'traffic_cone'])]
common_heads = dict(reg=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))

data_config = {'cams': ['CAM_FRONT_LEFT', 'CAM_FRONT', 'CAM_FRONT_RIGHT', 'CAM_BACK_LEFT', 'CAM_BACK', 'CAM_BACK_RIGHT'],
               'Ncams': 6, 'input_size': (256, 704), 'src_size': (900, 1600)}
grid_config = {'xbound': [-51.2, 51.2, 0.8], 'ybound': [-51.2, 51.2, 0.8],
               'zbound': [-10.0, 10.0, 20.0], 'dbound': [1.0, 60.0, 1.0]}
voxel_size = [0.1, 0.1, 0.2]
numC_Trans = 64

teacher_voxel_size = [0.2, 0.2, 8]
teacher = dict(model=dict(
    type='DynamicCenterPoint',
    pts_voxel_layer=dict(max_num_points=-1, voxel_size=teacher_voxel_size, max_voxels=(-1, -1),
                         point_cloud_range=point_cloud_range),
    pts_voxel_encoder=dict(type='DynamicPillarFeatureNet', in_channels=5, feat_channels=[64], with_distance=False,
                           voxel_size=(0.2, 0.2, 8), point_cloud_range=point_cloud_range,
                           norm_cfg=dict(type='BN1d', eps=1e-3, momentum=0.01)),
    pts_middle_encoder=dict(type='PointPillarsScatter', in_channels=64, output_shape=(512, 512)),
    pts_backbone=dict(type='SECOND', in_channels=64, out_channels=[64, 128, 256], layer_nums=[3, 5, 5],
                      layer_strides=[2, 2, 2], norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01),
                      conv_cfg=dict(type='Conv2d', bias=False)),
    pts_neck=dict(type='SECONDFPN', in_channels=[64, 128, 256], out_channels=[128, 128, 128],
                  upsample_strides=[0.5, 1, 2], norm_cfg=dict(type='BN', eps=1e-3, momentum=0.01),
                  upsample_cfg=dict(type='deconv', bias=False), use_conv_for_no_stride=True),
    pts_bbox_head=dict(
        type='CenterHead', in_channels=384, tasks=tasks, common_heads=common_heads, share_conv_channel=64,
        bbox_coder=dict(type='CenterPointBBoxCoder', post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                        max_num=500, score_threshold=0.1, out_size_factor=4, voxel_size=teacher_voxel_size[:2],
                        code_size=9, pc_range=point_cloud_range[:2]),
        separate_head=dict(type='SeparateHead', init_bias=-2.19, final_kernel=3),
        loss_cls=dict(type='GaussianFocalLoss', reduction='mean'),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25), norm_bbox=True),
    train_cfg=dict(pts=dict(grid_size=[512, 512, 1], voxel_size=teacher_voxel_size, out_size_factor=4, dense_reg=1,
                            gaussian_overlap=0.1, max_objs=500, min_radius=2,
                            code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
                            point_cloud_range=point_cloud_range)),
    test_cfg=dict(pts=dict(post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_per_img=500,
                           max_pool_nms=False, min_radius=[4, 12, 10, 1, 0.85, 0.175], score_threshold=0.1,
                           pc_range=point_cloud_range[:2], out_size_factor=4, voxel_size=teacher_voxel_size[:2],
                           nms_type='rotate', pre_max_size=1000, post_max_size=83, nms_thr=0.2))))

model = dict(
    type='BEVDepth4DDistill',
    teacher_config=teacher,
    teacher_ckpt=None,
    self_ckpt=None,
    inherit_head=True,
    distill_type='fgd',
    distill_params=dict(
        student_channels=[256, 512, 256], teacher_channels=[128, 256, 384],
        spatial_t=0.5, spatial_student_ratio=1.0, channel_t=0.5,
        fg_feat_loss_weights=[6e-3], bg_feat_loss_weights=[4e-2], channel_loss_weights=[0.25],
        spatial_loss_weights=[2.5e-3],
        adaptation_type=['upsample_3layer', 'upsample_3layer', '1x1conv'],
        student_adaptation_params=dict(kernel_size=1, stride=1, upsample_factor=4),
        teacher_adaptation_type='identity', teacher_adaptation_params=dict(kernel_size=4, stride=4),
        spatial_attentions=['teacher_student'],
        feat_criterion=dict(type='MSELoss', reduction='none'),
        spatial_criterion=dict(type='L1Loss', reduction='none'),
        channel_criterion=dict(type='L1Loss', reduction='none'),
        transpose_mask=False, foreground_mask='gt', background_mask='logical_not', scale_mask='combine_gt',
        spatial_mask=True, channel_mask=False,
        student_feat_pos=['backbone1', 'backbone2', 'head'], teacher_feat_pos=['backbone1', 'backbone2', 'head'],
        two_stage_epoch=-1, affinity_weights=[0], affinity_mode='none',
        affinity_criterion=dict(type='SmoothL1Loss'), affinity_split=1, non_empty_weight=0,
        output_threshold=0.1, groundtruth_threshold=None, fp_as_foreground=['none', 'none', 'teacher'],
        fp_weight=6e-2, fp_epoch=0, multi_scale_epoch=-1, fp_scale_mode='average', gauss_fg_weight=-1e10,
        context_length=0, context_weight=0),
    aligned=True, detach=True, before=True,
    img_backbone=dict(pretrained=None, type='ResNet', depth=50, num_stages=4, out_indices=(2, 3), frozen_stages=-1,
                      norm_cfg=dict(type='BN', requires_grad=True), norm_eval=False, with_cp=False, style='pytorch'),
    img_neck=dict(type='FPNForBEVDet', in_channels=[1024, 2048], out_channels=512, num_outs=1, start_level=0, out_ids=[0]),
    img_view_transformer=dict(type='ViewTransformerLSSBEVDepth', loss_depth_weight=100.0, grid_config=grid_config,
                              data_config=data_config, numC_Trans=numC_Trans,
                              extra_depth_net=dict(type='ResNetForBEVDet', numC_input=256, num_layer=[3],
                                                   num_channels=[256], stride=[1])),
    img_bev_encoder_backbone=dict(type='ResNetForBEVDet', numC_input=128, num_channels=[128, 256, 512]),
    img_bev_encoder_neck=dict(type='FPN_LSS', in_channels=numC_Trans * 8 + numC_Trans * 2, out_channels=256,
                              extra_norm_act=True),
    pre_process=dict(type='ResNetForBEVDet', numC_input=numC_Trans, num_layer=[2], num_channels=[64], stride=[1],
                     backbone_output_ids=[0]),
    pts_bbox_head=dict(
        type='CenterHead', task_specific=True, in_channels=256, tasks=tasks, common_heads=common_heads,
        share_conv_channel=64,
        bbox_coder=dict(type='CenterPointBBoxCoder', pc_range=point_cloud_range[:2],
                        post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], max_num=500, score_threshold=0.1,
                        out_size_factor=8, voxel_size=voxel_size[:2], code_size=9),
        separate_head=dict(type='SeparateHead', init_bias=-2.19, final_kernel=3),
        loss_cls=dict(type='GaussianFocalLoss', reduction='mean'),
        loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25), norm_bbox=True),
    train_cfg=dict(pts=dict(point_cloud_range=point_cloud_range, grid_size=[1024, 1024, 40], voxel_size=voxel_size,
                            out_size_factor=8, dense_reg=1, gaussian_overlap=0.1, max_objs=500, min_radius=2,
                            code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])),
    test_cfg=dict(pts=dict(pc_range=point_cloud_range[:2], post_center_limit_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0],
                           max_per_img=500, max_pool_nms=False, min_radius=[4, 12, 10, 1, 0.85, 0.175],
                           score_threshold=0.1, out_size_factor=8, voxel_size=voxel_size[:2], pre_max_size=1000,
                           post_max_size=83,
                           nms_type=['rotate', 'rotate', 'rotate', 'circle', 'rotate', 'rotate'],
                           nms_thr=[0.2, 0.2, 0.2, 0.2, 0.2, 0.5],
                           nms_rescale_factor=[1.0, [0.7, 0.7], [0.4, 0.55], 1.1, [1.0, 1.0], [4.5, 9.0]])))

# optimisation recipe (configs/.../to_bevdepth4d_r50.py optimizer + the script's overrides)
optimizer = dict(type='AdamW', lr=2e-4, weight_decay=0.01)
optimizer_config = dict(grad_clip=dict(max_norm=5, norm_type=2))
data = dict(samples_per_gpu=8)
