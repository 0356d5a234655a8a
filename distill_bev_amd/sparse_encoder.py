"""Voxel-teacher encoders of the MVP / LidarFormer configs -- registry mirrors of
  ``mmdet3d/ops/sparse_block.py``            SparseBasicBlock :66-121, make_sparse_convmodule :124-186
  ``mmdet3d/models/middle_encoders/sparse_encoder.py``   SparseEncoder :11-204
  ``mmdet3d/models/voxel_encoders/dynamic_voxel_encoder.py``  voxelization :8-18, voxelization_virtual :19-68,
                                                              DynamicVoxelEncoder :70-102
on the sparse-convolution kernels (spconv.py) and the dynamic-scatter kernels (voxel.py).  State-dict keys follow the
reference (conv_input.0.weight, encoder_layers.encoder_layer1.0.conv1.weight, ...bn1..., conv_out.0.weight)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import spconv
from .registry import MODELS, build_conv_layer, build_norm_layer
from .voxel import dynamic_scatter_prepare, dynamic_scatter_reduce


class SparseBasicBlock(spconv.SparseModule):
    """mmdet BasicBlock (conv1-bn1-relu-conv2-bn2, + identity, relu) over sparse tensors; expansion 1."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        norm_cfg = norm_cfg or dict(type="BN1d")
        self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
        self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1, dilation=1, bias=False)
        self.add_module(self.norm1_name, norm1)
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.add_module(self.norm2_name, norm2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    @property
    def norm1(self):
        return getattr(self, self.norm1_name)

    @property
    def norm2(self):
        return getattr(self, self.norm2_name)

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2, f"x.features.dim()={x.features.dim()}"
        if spconv.fusable_norm(self.conv1, self.norm1) and spconv.fusable_norm(self.conv2, self.norm2):
            # inference: both norm / add / relu tails run in the convolutions' epilogues
            if self.downsample is not None:
                identity = self.downsample(x)
            out = self.conv1(x, fused=(self.norm1, None, True))
            return self.conv2(out, fused=(self.norm2, identity, True))
        out = self.conv1(x)
        out.features = self.relu(self.norm1(out.features))
        out = self.conv2(out)
        out.features = self.norm2(out.features)
        if self.downsample is not None:
            identity = self.downsample(x)
        out.features = self.relu(out.features + identity)
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0, conv_type="SubMConv3d",
                           norm_cfg=None, order=("conv", "norm", "act")):
    """sparse_block.py:124-186"""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {"conv", "norm", "act"} == {"conv", "norm", "act"}
    conv_cfg = dict(type=conv_type, indice_key=indice_key)
    layers = []
    for layer in order:
        if layer == "conv":
            if conv_type not in ("SparseInverseConv3d", "SparseInverseConv2d", "SparseInverseConv1d"):
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, stride=stride,
                                               padding=padding, bias=False))
            else:
                layers.append(build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size, bias=False))
        elif layer == "norm":
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == "act":
            layers.append(nn.ReLU(inplace=True))
    return spconv.SparseSequential(*layers)


@MODELS.register_module()
class SparseEncoder(nn.Module):
    """sparse_encoder.py:11-204"""

    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module"):
        super().__init__()
        assert block_type in ["conv_module", "basicblock"]
        order = tuple(order)
        self.sparse_shape, self.in_channels, self.order = sparse_shape, in_channels, order
        self.base_channels, self.output_channels = base_channels, output_channels
        self.encoder_channels, self.encoder_paddings = encoder_channels, encoder_paddings
        self.stage_num = len(self.encoder_channels)
        assert len(order) == 3 and set(order) == {"conv", "norm", "act"}
        if order[0] != "conv":      # pre activate
            self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key="subm1", conv_type="SubMConv3d", order=("conv",))
        else:
            self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                     indice_key="subm1", conv_type="SubMConv3d")
        encoder_out_channels = self.make_encoder_layers(make_sparse_convmodule, norm_cfg, base_channels, block_type=block_type)
        self.conv_out = make_sparse_convmodule(encoder_out_channels, output_channels, kernel_size=(3, 1, 1), stride=(2, 1, 1),
                                               norm_cfg=norm_cfg, padding=0, indice_key="spconv_down2",
                                               conv_type="SparseConv3d")

    def forward(self, voxel_features, coors, batch_size):
        """voxel_features [N, C], coors int [N, 4] = (batch, z, y, x) -> [B, C * D, H, W] dense BEV map."""
        coors = coors.int()
        x = spconv.SparseConvTensor(voxel_features, coors, self.sparse_shape, batch_size)
        x = self.conv_input(x)
        encode_features = []
        for encoder_layer in self.encoder_layers:
            x = encoder_layer(x)
            encode_features.append(x)
        out = self.conv_out(encode_features[-1])
        spatial_features = out.dense()
        N, C, D, H, W = spatial_features.shape
        return spatial_features.view(N, C * D, H, W)

    def make_encoder_layers(self, make_block, norm_cfg, in_channels, block_type="conv_module",
                            conv_cfg=dict(type="SubMConv3d")):
        assert block_type in ["conv_module", "basicblock"]
        self.encoder_layers = spconv.SparseSequential()
        for i, blocks in enumerate(self.encoder_channels):
            blocks_list = []
            for j, out_channels in enumerate(tuple(blocks)):
                padding = tuple(self.encoder_paddings[i])[j]
                if i != 0 and j == 0 and block_type == "conv_module":
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2, padding=padding,
                                                  indice_key=f"spconv{i + 1}", conv_type="SparseConv3d"))
                elif block_type == "basicblock":
                    if j == len(blocks) - 1 and i != len(self.encoder_channels) - 1:
                        blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, stride=2,
                                                      padding=padding, indice_key=f"spconv{i + 1}",
                                                      conv_type="SparseConv3d"))
                    else:
                        blocks_list.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg, conv_cfg=conv_cfg))
                else:
                    blocks_list.append(make_block(in_channels, out_channels, 3, norm_cfg=norm_cfg, padding=padding,
                                                  indice_key=f"subm{i + 1}", conv_type="SubMConv3d"))
                in_channels = out_channels
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", spconv.SparseSequential(*blocks_list))
        return out_channels


# ---- dynamic_voxel_encoder.py ---------------------------------------------------------------------------------------------
def _host3(t):
    """range / size triple on the host (the ABI takes them by value; a device tensor costs one read-back)"""
    return [float(v) for v in (t.tolist() if torch.is_tensor(t) else t)]


def _group_by_voxel(points, pc_range, voxel_size, virtual):
    """Range test + (z, y, x) voxel of every point (dbev_range_voxel_coords), then the dynamic-scatter grouping: voxels in the
    lexicographic order of `coords.unique(dim=0)` (dynamic_voxel_encoder.py:14,58), the points of a voxel in ascending id.
    Points on the upper range border fall into cell index `shape` (the reference keeps them): the grid has one more cell
    per axis than the encoder's `shape`."""
    dev = L.require_cuda(points)
    points = points.contiguous()
    assert points.dtype == torch.float32 and points.dim() == 2
    rng, vs = _host3(pc_range), _host3(voxel_size)
    n, F = points.shape
    coors = torch.empty((n, 3), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_range_voxel_coords", L.ptr(points), n, F, L.host_floats(rng), L.host_floats(vs), int(virtual),
               L.ptr(coors), L.stream_ptr(dev))
    # largest index = trunc(fp32((max - min) / size)): fp32 subtraction and division are monotonic
    grid = tuple(int((np.float32(rng[3 + k]) - np.float32(rng[k])) / np.float32(vs[k])) + 1 for k in (2, 1, 0))
    return points, dynamic_scatter_prepare(coors, grid)


def voxelization(points, pc_range, voxel_size):
    """dynamic_voxel_encoder.py:8-17: per-voxel mean of the rows inside the range -> (voxels [M, F], coords [M, 3] (z, y, x))"""
    points, prep = _group_by_voxel(points, pc_range, voxel_size, False)
    return dynamic_scatter_reduce(points, prep, "mean"), prep["out_coors"].long()


def voxelization_virtual(points, pc_range, voxel_size):
    """dynamic_voxel_encoder.py:19-68 (MVP virtual points; column -2 = 1 real / 0 painted / -1 virtual): 23 columns per
    voxel -- real points' (x y z i t score) means in 0..5, painted / virtual points' 15 columns + tag + painted flag in 6..22,
    voxels holding both kinds rescaled by the real fraction -- in one pass over the voxel's rows (dbev_virtual_voxel_reduce),
    no padded copy of the cloud."""
    points, prep = _group_by_voxel(points, pc_range, voxel_size, True)
    dev, M = points.device, prep["M"]
    voxels = torch.empty((M, 23), dtype=torch.float32, device=dev)
    if M > 0:
        with torch.cuda.device(dev):
            L.call("dbev_virtual_voxel_reduce", L.ptr(points), L.ptr(prep["vstart"]), L.ptr(prep["vlist"]), L.ptr(voxels), M,
                   L.stream_ptr(dev))
    return voxels, prep["out_coors"].long()


@MODELS.register_module()
class DynamicVoxelEncoder(nn.Module):
    """dynamic_voxel_encoder.py:70-102"""

    def __init__(self, pc_range, voxel_size, virtual=False):
        super().__init__()
        self.pc_range = torch.tensor(pc_range)
        self.voxel_size = torch.tensor(voxel_size)
        self.shape = torch.round((self.pc_range[3:] - self.pc_range[:3]) / self.voxel_size)
        self.shape_np = self.shape.numpy().astype(np.int32)
        self.virtual = virtual

    @torch.no_grad()
    def forward(self, points):
        coors, voxels = [], []
        for res in points:
            fn = voxelization_virtual if self.virtual else voxelization
            voxel, coor = fn(res, self.pc_range, self.voxel_size)  # host triples: passed by value to the kernels
            voxels.append(voxel)
            coors.append(coor)
        coors_batch = torch.cat([F.pad(c, (1, 0), mode="constant", value=i) for i, c in enumerate(coors)], dim=0)
        return torch.cat(voxels, dim=0), coors_batch, self.shape_np


# ---- the voxel teachers of configs/teacher_transformer/{lidarformer,mvpformer}.py: feature-extraction path ---------------
@MODELS.register_module()
class HardSimpleVFE(nn.Module):
    """voxel_encoders/voxel_encoder.py:14-45: mean of the points of a (hard) voxel."""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features

    def forward(self, features, num_points, coors):
        return (features[:, :, :self.num_features].sum(dim=1) / num_points.type_as(features).view(-1, 1)).contiguous()


@MODELS.register_module()
class FPN(nn.Module):
    """mmdet==2.24.0 FPN (un-vendored; configured at mvpformer.py:60-67 with norm_cfg BN2d / act_cfg ReLU, 2 inputs -> 4
    outputs, and as the BEVFormer image neck with add_extra_convs='on_output', 3 inputs -> 4 outputs): lateral 1x1
    ConvModules, top-down nearest upsampling, 3x3 fpn ConvModules; extra levels by stride-2 max-pooling or by stride-2 3x3
    convolutions on the input / lateral / output of the last level."""

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__()
        from .registry import ConvModule
        assert isinstance(add_extra_convs, (str, bool))
        if add_extra_convs is True:
            add_extra_convs = "on_input"
        assert add_extra_convs in (False, "on_input", "on_lateral", "on_output")
        self.add_extra_convs, self.relu_before_extra_convs = add_extra_convs, relu_before_extra_convs
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.start_level = start_level
        self.backbone_end_level = len(in_channels) if end_level == -1 else end_level
        self.upsample_cfg = dict(upsample_cfg)
        self.lateral_convs, self.fpn_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg if not no_norm_on_lateral else None, act_cfg=act_cfg,
                                                 inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                             act_cfg=act_cfg, inplace=False))
        extra = num_outs - self.backbone_end_level + self.start_level
        if self.add_extra_convs and extra >= 1:
            for i in range(extra):
                cin = in_channels[self.backbone_end_level - 1] if (i == 0 and self.add_extra_convs == "on_input") else out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                                 act_cfg=act_cfg, inplace=False))

    def init_weights(self):
        for m in self.modules():                 # init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        laterals = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        used = len(laterals)
        for i in range(used - 1, 0, -1):
            laterals[i - 1] = laterals[i - 1] + F.interpolate(laterals[i], size=laterals[i - 1].shape[2:], **self.upsample_cfg)
        outs = [self.fpn_convs[i](laterals[i]) for i in range(used)]
        if self.num_outs > len(outs):
            if not self.add_extra_convs:
                for _ in range(self.num_outs - used):
                    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
            else:
                src = {"on_input": inputs[self.backbone_end_level - 1], "on_lateral": laterals[-1], "on_output": outs[-1]}
                outs.append(self.fpn_convs[used](src[self.add_extra_convs]))
                for i in range(used + 1, self.num_outs):
                    outs.append(self.fpn_convs[i](F.relu(outs[-1]) if self.relu_before_extra_convs else outs[-1]))
        return tuple(outs)


class _VoxelTeacher(nn.Module):
    """Shared part of LidarFormer (lidarformer.py:11-53) and MVPFormer (mvpformer.py:11-49): voxel encoder -> SparseEncoder ->
    SECOND -> FPN -> DGCNN3DHead (whose ``bev_embed`` the BEVFormer distillation reads)."""

    def __init__(self, pts_voxel_layer=None, pts_voxel_encoder=None, pts_middle_encoder=None, pts_backbone=None, pts_neck=None,
                 pts_bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None, **unused):
        super().__init__()
        from .registry import build_backbone, build_head, build_neck
        from .voxel import Voxelization
        if pts_voxel_layer:
            self.pts_voxel_layer = Voxelization(**pts_voxel_layer)
        self.pts_voxel_encoder = MODELS.build(pts_voxel_encoder)
        self.pts_middle_encoder = MODELS.build(pts_middle_encoder)
        self.pts_backbone = build_backbone(pts_backbone)
        self.pts_neck = build_neck(pts_neck) if pts_neck is not None else None
        if pts_bbox_head:                                    # mvx_two_stage.py:60-66
            from . import detr_head  # noqa: F401
            head = dict(pts_bbox_head)
            head.update(train_cfg=train_cfg["pts"] if train_cfg else None, test_cfg=test_cfg["pts"] if test_cfg else None)
            self.pts_bbox_head = build_head(head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def init_weights(self):
        head = getattr(self, "pts_bbox_head", None)
        if head is not None:
            head.init_weights()

    def extract_feat(self, points, img=None, img_metas=None):
        """mvpformer.py:51-55 -> (img_feats = None, pts_feats)"""
        return None, self.extract_pts_feat(points, None, img_metas)

    def forward_pts_train(self, pts_feats, gt_bboxes_3d, gt_labels_3d, img_metas=None, gt_bboxes_ignore=None):
        """mvpformer.py:57-81"""
        return self.pts_bbox_head.loss(gt_bboxes_3d, gt_labels_3d, self.pts_bbox_head(pts_feats))

    def forward_train(self, points=None, img_metas=None, gt_bboxes_3d=None, gt_labels_3d=None, **kwargs):
        return dict(self.forward_pts_train(self.extract_feat(points, None, img_metas)[1], gt_bboxes_3d, gt_labels_3d, img_metas))

    def simple_test_pts(self, x, img_metas, rescale=False):
        """mvpformer.py:84-103 -> per sample (boxes, scores, labels)"""
        return self.pts_bbox_head.get_bboxes(self.pts_bbox_head(x), img_metas, rescale=rescale)

    def _dense(self, voxel_features, coors, batch_size):
        x = self.pts_middle_encoder(voxel_features, coors, batch_size)
        x = self.pts_backbone(x)
        return self.pts_neck(x) if self.pts_neck is not None else x


@MODELS.register_module()
class MVPFormer(_VoxelTeacher):
    def extract_pts_feat(self, pts, img_feats=None, img_metas=None):
        voxel_features, coors, _ = self.pts_voxel_encoder(pts)
        return self._dense(voxel_features, coors, len(pts))


@MODELS.register_module()
class LidarFormer(_VoxelTeacher):
    @torch.no_grad()
    def voxelize(self, points):
        """mvx_two_stage.py:217-242"""
        voxels, coors, num_points = [], [], []
        for res in points:
            v, c, n = self.pts_voxel_layer(res)
            voxels.append(v); coors.append(c); num_points.append(n)
        coors_batch = torch.cat([F.pad(c, (1, 0), mode="constant", value=i) for i, c in enumerate(coors)], dim=0)
        return torch.cat(voxels, dim=0), torch.cat(num_points, dim=0), coors_batch

    def extract_pts_feat(self, pts, img_feats=None, img_metas=None):
        voxels, num_points, coors = self.voxelize(pts)
        voxel_features = self.pts_voxel_encoder(voxels, num_points, coors)
        return self._dense(voxel_features, coors, len(pts))
