"""BEVFormer / deformable-DETR transformer stack of the MVP -> BEVFormer distillation step (BASELINE configs[4]).

Mirrors, type name by type name and key by key, what the reference's configs build
(``configs/lidar2camera_bev_distillation/teacher_to_bevformer/*.py``, ``configs/teacher_transformer/*.py``):

* the reference's own modules ``mmdet3d/models/transformer_modules/``: ``TemporalSelfAttention``
  (temporal_self_attention.py:25-264), ``SpatialCrossAttention`` / ``MSDeformableAttention3D``
  (spatial_cross_attention.py:32-397), ``CustomMSDeformableAttention`` / ``DetectionTransformerDecoder`` (decoder.py:46-344),
  ``BEVFormerEncoder`` / ``BEVFormerLayer`` (encoder.py:29-400), ``MyCustomBaseTransformerLayer``
  (custom_base_transformer_layer.py:35-259), ``PerceptionTransformer`` (perception_transformer.py:22-280),
  ``DeformableDetrTransformerDistill`` (detr_transformer.py:24-355);
* the un-vendored mmcv 1.x / mmdet 2.24 bricks those configs name: ``FFN``, ``MultiheadAttention``, ``BaseTransformerLayer``,
  ``DetrTransformerDecoderLayer``, ``TransformerLayerSequence``, ``DetrTransformerEncoder``, ``MultiScaleDeformableAttention``,
  ``SinePositionalEncoding``, ``LearnedPositionalEncoding`` -- parameter names follow those packages so their checkpoints load.

Every deformable attention samples through ``dbev_msda_forward / dbev_msda_backward`` (csrc/msda.hip).  What differs from
the reference by design: the camera re-batching of SpatialCrossAttention is planned ONCE per encoder pass on the device
(one index sort + one host read of the six lengths, instead of bs x 6 ``nonzero`` calls in each of the six layers) and
applied with gathers / per-camera ``index_add_`` (unique indices: deterministic); ``point_sampling`` multiplies the
lidar2img matrices as one batched GEMM per (sample, camera) instead of D*B*6*Q broadcast 4x4 ``matmul``s.
"""
import copy
import math
import warnings

import numpy as np
import torch
import torch.nn as nn

from .msda import level_tensors, multi_scale_deformable_attn
from . import _lib as L
from .registry import MODELS, build_activation_layer, build_norm_layer

build_attention = build_feedforward_network = build_positional_encoding = MODELS.build
build_transformer_layer = build_transformer_layer_sequence = build_transformer = MODELS.build


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    """mmcv.cnn.xavier_init: a no-op on containers (no ``weight`` attribute), as in mmcv."""
    if getattr(module, "weight", None) is not None:
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def inverse_sigmoid(x, eps=1e-5):
    """decoder.py:30-43"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


# ---- mmcv bricks -------------------------------------------------------------------------------------------------------
@MODELS.register_module()
class FFN(nn.Module):
    """mmcv.cnn.bricks.transformer.FFN: (Linear, act, Dropout) x (num_fcs - 1), Linear, Dropout, + identity."""

    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type="ReLU", inplace=True),
                 ffn_drop=0.0, dropout_layer=None, add_identity=True, init_cfg=None, **kwargs):
        super().__init__()
        assert num_fcs >= 2
        ffn_drop = kwargs.pop("dropout", ffn_drop)
        add_identity = kwargs.pop("add_residual", add_identity)
        self.embed_dims, self.feedforward_channels, self.num_fcs = embed_dims, feedforward_channels, num_fcs
        layers, cin = [], embed_dims
        for _ in range(num_fcs - 1):
            layers.append(nn.Sequential(nn.Linear(cin, feedforward_channels), build_activation_layer(act_cfg),
                                        nn.Dropout(ffn_drop)))
            cin = feedforward_channels
        layers += [nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = nn.Sequential(*layers)
        self.dropout_layer = _build_dropout(dropout_layer)
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.dropout_layer(self.layers(x))
        if not self.add_identity:
            return out
        return (x if identity is None else identity) + out


def _build_dropout(cfg):
    if not cfg:
        return nn.Identity()
    cfg = dict(cfg)
    typ = cfg.pop("type", "Dropout")
    assert typ == "Dropout", "only the plain Dropout layer is configured by the reference"
    return nn.Dropout(cfg.get("drop_prob", cfg.get("p", 0.5)))


@MODELS.register_module()
class MultiheadAttention(nn.Module):
    """mmcv.cnn.bricks.transformer.MultiheadAttention: nn.MultiheadAttention + positional adds + identity."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0, dropout_layer=dict(type="Dropout", drop_prob=0.0),
                 init_cfg=None, batch_first=False, **kwargs):
        super().__init__()
        if "dropout" in kwargs:                      # the deprecated spelling the reference's configs use
            attn_drop = kwargs.pop("dropout")
            dropout_layer = dict(type="Dropout", drop_prob=attn_drop)
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = _build_dropout(dropout_layer)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if key_pos is None and query_pos is not None and query_pos.shape == key.shape:
            key_pos = query_pos
        if query_pos is not None:
            query = query + query_pos
        if key_pos is not None:
            key = key + key_pos
        if self.batch_first:
            query, key, value = query.transpose(0, 1), key.transpose(0, 1), value.transpose(0, 1)
        out = self.attn(query=query, key=key, value=value, attn_mask=attn_mask, key_padding_mask=key_padding_mask)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


@MODELS.register_module()
class BaseTransformerLayer(nn.Module):
    """mmcv BaseTransformerLayer (batch_first False by default) == custom_base_transformer_layer.py:35-259 (True)."""
    default_batch_first = False

    def __init__(self, attn_cfgs=None,
                 ffn_cfgs=dict(type="FFN", embed_dims=256, feedforward_channels=1024, num_fcs=2, ffn_drop=0.0,
                               act_cfg=dict(type="ReLU", inplace=True)),
                 operation_order=None, norm_cfg=dict(type="LN"), init_cfg=None, batch_first=None, **kwargs):
        super().__init__()
        ffn_cfgs = copy.deepcopy(dict(ffn_cfgs))
        for old, new in (("feedforward_channels", "feedforward_channels"), ("ffn_dropout", "ffn_drop"), ("ffn_num_fcs", "num_fcs")):
            if old in kwargs:
                ffn_cfgs[new] = kwargs[old]
        self.batch_first = self.default_batch_first if batch_first is None else batch_first
        assert set(operation_order) <= {"self_attn", "norm", "ffn", "cross_attn"}
        num_attn = operation_order.count("self_attn") + operation_order.count("cross_attn")
        if isinstance(attn_cfgs, dict):
            attn_cfgs = [copy.deepcopy(attn_cfgs) for _ in range(num_attn)]
        else:
            assert num_attn == len(attn_cfgs)
            attn_cfgs = [copy.deepcopy(dict(c)) for c in attn_cfgs]
        self.num_attn, self.operation_order, self.norm_cfg = num_attn, tuple(operation_order), norm_cfg
        self.pre_norm = operation_order[0] == "norm"
        self.attentions = nn.ModuleList()
        for name, cfg in zip([o for o in operation_order if o in ("self_attn", "cross_attn")], attn_cfgs):
            if "batch_first" in cfg:
                assert self.batch_first == cfg["batch_first"]
            else:
                cfg["batch_first"] = self.batch_first
            attention = build_attention(cfg)
            attention.operation_name = name
            self.attentions.append(attention)
        self.embed_dims = self.attentions[0].embed_dims
        self.ffns = nn.ModuleList()
        for _ in range(operation_order.count("ffn")):
            cfg = copy.deepcopy(ffn_cfgs)
            cfg.setdefault("embed_dims", self.embed_dims)
            assert cfg["embed_dims"] == self.embed_dims
            self.ffns.append(build_feedforward_network(cfg))
        self.norms = nn.ModuleList(build_norm_layer(norm_cfg, self.embed_dims)[1]
                                   for _ in range(operation_order.count("norm")))

    def _masks(self, attn_masks):
        if attn_masks is None:
            return [None] * self.num_attn
        if isinstance(attn_masks, torch.Tensor):
            return [attn_masks.clone() for _ in range(self.num_attn)]
        assert len(attn_masks) == self.num_attn
        return attn_masks

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, **kwargs):
        attn_masks = self._masks(attn_masks)
        ni = ai = fi = 0
        identity = query
        for op in self.operation_order:
            if op == "self_attn":
                query = self.attentions[ai](query, query, query, identity if self.pre_norm else None, query_pos=query_pos,
                                            key_pos=query_pos, attn_mask=attn_masks[ai],
                                            key_padding_mask=query_key_padding_mask, **kwargs)
                ai += 1
                identity = query
            elif op == "norm":
                query = self.norms[ni](query)
                ni += 1
            elif op == "cross_attn":
                query = self.attentions[ai](query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                                            key_pos=key_pos, attn_mask=attn_masks[ai], key_padding_mask=key_padding_mask,
                                            **kwargs)
                ai += 1
                identity = query
            else:
                query = self.ffns[fi](query, identity if self.pre_norm else None)
                fi += 1
        return query


@MODELS.register_module()
class MyCustomBaseTransformerLayer(BaseTransformerLayer):
    default_batch_first = True


@MODELS.register_module()
class DetrTransformerDecoderLayer(BaseTransformerLayer):
    """mmdet.models.utils.transformer.DetrTransformerDecoderLayer"""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels, ffn_dropout=ffn_dropout,
                         operation_order=operation_order, norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6 and set(operation_order) == {"self_attn", "norm", "cross_attn", "ffn"}


@MODELS.register_module()
class TransformerLayerSequence(nn.Module):
    """mmcv TransformerLayerSequence: ``num_layers`` layers built from one (or a list of) layer config."""

    def __init__(self, transformerlayers=None, num_layers=None, init_cfg=None):
        super().__init__()
        if isinstance(transformerlayers, dict):
            transformerlayers = [copy.deepcopy(transformerlayers) for _ in range(num_layers)]
        assert isinstance(transformerlayers, (list, tuple)) and len(transformerlayers) == num_layers
        self.num_layers = num_layers
        self.layers = nn.ModuleList(build_transformer_layer(cfg) for cfg in transformerlayers)
        self.embed_dims = self.layers[0].embed_dims
        self.pre_norm = self.layers[0].pre_norm

    def forward(self, query, key, value, query_pos=None, key_pos=None, attn_masks=None, query_key_padding_mask=None,
                key_padding_mask=None, **kwargs):
        for layer in self.layers:
            query = layer(query, key, value, query_pos=query_pos, key_pos=key_pos, attn_masks=attn_masks,
                          query_key_padding_mask=query_key_padding_mask, key_padding_mask=key_padding_mask, **kwargs)
        return query


@MODELS.register_module()
class DetrTransformerEncoder(TransformerLayerSequence):
    """mmdet DetrTransformerEncoder: a final LayerNorm only for pre-norm layers."""

    def __init__(self, *args, post_norm_cfg=dict(type="LN"), **kwargs):
        super().__init__(*args, **kwargs)
        self.post_norm = build_norm_layer(post_norm_cfg, self.embed_dims)[1] if (post_norm_cfg is not None and self.pre_norm) \
            else None

    def forward(self, *args, **kwargs):
        x = super().forward(*args, **kwargs)
        return x if self.post_norm is None else self.post_norm(x)


# ---- positional encodings (mmdet.models.utils.positional_encoding) ---------------------------------------------------------
@MODELS.register_module()
class SinePositionalEncoding(nn.Module):
    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6, offset=0.0, init_cfg=None):
        super().__init__()
        self.num_feats, self.temperature, self.normalize, self.scale, self.eps, self.offset = \
            num_feats, temperature, normalize, scale, eps, offset

    def forward(self, mask):
        """mask [bs, h, w] (non-zero = ignored) -> [bs, 2 * num_feats, h, w]"""
        not_mask = 1 - mask.to(torch.int)
        y = not_mask.cumsum(1, dtype=torch.float32)
        x = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y = (y + self.offset) / (y[:, -1:, :] + self.eps) * self.scale
            x = (x + self.offset) / (x[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_feats)
        B, H, W = mask.size()
        px, py = x[:, :, :, None] / dim_t, y[:, :, :, None] / dim_t
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


@MODELS.register_module()
class LearnedPositionalEncoding(nn.Module):
    def __init__(self, num_feats, row_num_embed=50, col_num_embed=50, init_cfg=None):
        super().__init__()
        self.row_embed = nn.Embedding(row_num_embed, num_feats)
        self.col_embed = nn.Embedding(col_num_embed, num_feats)
        self.num_feats, self.row_num_embed, self.col_num_embed = num_feats, row_num_embed, col_num_embed
        nn.init.uniform_(self.row_embed.weight)
        nn.init.uniform_(self.col_embed.weight)

    def forward(self, mask):
        h, w = mask.shape[-2:]
        xe = self.col_embed(torch.arange(w, device=mask.device))
        ye = self.row_embed(torch.arange(h, device=mask.device))
        pos = torch.cat((xe.unsqueeze(0).expand(h, w, -1), ye.unsqueeze(1).expand(h, w, -1)), dim=-1)
        return pos.permute(2, 0, 1).unsqueeze(0).repeat(mask.shape[0], 1, 1, 1)


# ---- deformable attention ----------------------------------------------------------------------------------------------
def _ring_offsets(num_heads, groups, num_points):
    """The deformable-DETR offset prior: head h looks along angle 2*pi*h/num_heads, point i at distance i + 1."""
    thetas = torch.arange(num_heads, dtype=torch.float32) * (2.0 * math.pi / num_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(num_heads, 1, 1, 2).repeat(1, groups, num_points, 1)
    for i in range(num_points):
        grid[:, :, i, :] *= i + 1
    return grid.view(-1)


def _num_keys(spatial_shapes):
    """sum of H * W over the levels -- from the host copy the level tensors of `level_tensors` carry (no device read-back)"""
    hw = getattr(spatial_shapes, "_dbev_host", None)
    if hw is None:
        return int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum())
    return sum(hw[i] * hw[i + 1] for i in range(0, len(hw), 2))


class _DeformableAttention(nn.Module):
    """Shared construction of the four deformable attentions: value / offset / weight projections and their init."""

    def _build(self, embed_dims, num_heads, num_levels, num_points, queue=1, output_proj=True):
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        d = embed_dims // num_heads
        if d & (d - 1):
            warnings.warn("the head dimension of a deformable attention should be a power of 2")
        self.embed_dims, self.num_heads, self.num_levels, self.num_points = embed_dims, num_heads, num_levels, num_points
        self._queue = queue
        self.sampling_offsets = nn.Linear(embed_dims * queue, queue * num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims * queue, queue * num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims) if output_proj else None
        self.init_weights()

    def init_weights(self):
        constant_init(self.sampling_offsets, 0.0)
        self.sampling_offsets.bias.data = _ring_offsets(self.num_heads, self.num_levels * self._queue, self.num_points)
        constant_init(self.attention_weights, val=0.0, bias=0.0)
        xavier_init(self.value_proj, distribution="uniform", bias=0.0)
        if self.output_proj is not None:
            xavier_init(self.output_proj, distribution="uniform", bias=0.0)
        self._is_init = True

    init_weight = init_weights

    @staticmethod
    def _locations(reference_points, offsets, spatial_shapes, num_points):
        """sampling locations [bs, Q, heads, levels, points, 2] from reference points [bs, Q, levels (or 1), 2 | 4]"""
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            return reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        if reference_points.shape[-1] == 4:
            return reference_points[:, :, None, :, None, :2] + offsets / num_points * reference_points[:, :, None, :, None, 2:] * 0.5
        raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")


@MODELS.register_module()
class MultiScaleDeformableAttention(_DeformableAttention):
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttention (the teacher's BEV encoder layers)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1, batch_first=False,
                 norm_cfg=None, init_cfg=None):
        super().__init__()
        self.norm_cfg, self.batch_first, self.im2col_step = norm_cfg, batch_first, im2col_step
        self.dropout = nn.Dropout(dropout)
        self._build(embed_dims, num_heads, num_levels, num_points)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        value = query if value is None else value
        identity = query if identity is None else identity
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nq, _ = query.shape
        _, nv, _ = value.shape
        assert _num_keys(spatial_shapes) == nv
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        offsets = self.sampling_offsets(query).view(bs, nq, self.num_heads, self.num_levels, self.num_points, 2)
        weights = self.attention_weights(query).view(bs, nq, self.num_heads, self.num_levels * self.num_points).softmax(-1)
        weights = weights.view(bs, nq, self.num_heads, self.num_levels, self.num_points)
        loc = self._locations(reference_points, offsets, spatial_shapes, self.num_points)
        out = multi_scale_deformable_attn(value, spatial_shapes, level_start_index, loc, weights, self.im2col_step)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


@MODELS.register_module()
class CustomMSDeformableAttention(MultiScaleDeformableAttention):
    """decoder.py:131-344: the decoder's cross attention over the BEV map (one level)."""


@MODELS.register_module()
class MSDeformableAttention3D(_DeformableAttention):
    """spatial_cross_attention.py:173-397: every BEV query owns ``num_Z_anchors`` reference points per camera image and
    samples ``num_points / num_Z_anchors`` offsets around each; no output projection (SpatialCrossAttention has it)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=8, im2col_step=64, dropout=0.1, batch_first=True,
                 norm_cfg=None, init_cfg=None):
        super().__init__()
        self.norm_cfg, self.batch_first, self.im2col_step = norm_cfg, batch_first, im2col_step
        self._build(embed_dims, num_heads, num_levels, num_points, output_proj=False)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, **kwargs):
        value = query if value is None else value
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nq, _ = query.shape
        _, nv, _ = value.shape
        assert _num_keys(spatial_shapes) == nv
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, nv, self.num_heads, -1)
        offsets = self.sampling_offsets(query).view(bs, nq, self.num_heads, self.num_levels, self.num_points, 2)
        weights = self.attention_weights(query).view(bs, nq, self.num_heads, self.num_levels * self.num_points).softmax(-1)
        weights = weights.view(bs, nq, self.num_heads, self.num_levels, self.num_points)
        assert reference_points.shape[-1] == 2
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        nz = reference_points.shape[2]
        assert self.num_points % nz == 0
        offsets = (offsets / normalizer[None, None, None, :, None, :]).view(bs, nq, self.num_heads, self.num_levels,
                                                                            self.num_points // nz, nz, 2)
        loc = (reference_points[:, :, None, None, None, :, :] + offsets).view(bs, nq, self.num_heads, self.num_levels,
                                                                              self.num_points, 2)
        out = multi_scale_deformable_attn(value, spatial_shapes, level_start_index, loc, weights, self.im2col_step)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return out


class CameraPlan(object):
    """Which BEV queries each camera sees, as padded index rows (spatial_cross_attention.py:132-151 builds the same lists
    with ``nonzero`` per camera, per layer).  ``index [num_cams, max_len]`` ascending query ids, ``valid`` the padding mask,
    ``lens`` host ints, ``count [bs, Q]`` cameras seeing a query (>= 1).  As in the reference the lists come from sample 0
    of the batch and are applied to every sample; ``count`` uses each sample's own mask."""

    def __init__(self, bev_mask):
        hit = bev_mask[:, 0].any(-1)                                   # [num_cams, Q]
        order = torch.sort(hit.to(torch.uint8), dim=1, descending=True, stable=True)[1]
        self.lens = [int(v) for v in hit.sum(1).tolist()]            # the one host read of the encoder pass
        self.max_len = max(self.lens)
        self.index = order[:, :self.max_len].contiguous()
        self.valid = torch.arange(self.max_len, device=hit.device)[None, :] < hit.sum(1, keepdim=True)
        seen = (bev_mask.sum(-1) > 0).permute(1, 2, 0).sum(-1)
        self.count = torch.clamp(seen, min=1.0)


@MODELS.register_module()
class SpatialCrossAttention(nn.Module):
    """spatial_cross_attention.py:32-170"""

    def __init__(self, embed_dims=256, num_cams=6, pc_range=None, dropout=0.1, init_cfg=None, batch_first=False,
                 deformable_attention=dict(type="MSDeformableAttention3D", embed_dims=256, num_levels=4), **kwargs):
        super().__init__()
        self.dropout = nn.Dropout(dropout)
        self.pc_range, self.embed_dims, self.num_cams, self.batch_first = pc_range, embed_dims, num_cams, batch_first
        self.deformable_attention = build_attention(deformable_attention)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weight()

    def init_weight(self):
        xavier_init(self.output_proj, distribution="uniform", bias=0.0)

    def forward(self, query, key, value, residual=None, query_pos=None, key_padding_mask=None, reference_points=None,
                spatial_shapes=None, reference_points_cam=None, bev_mask=None, level_start_index=None, flag="encoder",
                camera_plan=None, **kwargs):
        key = query if key is None else key
        value = key if value is None else value
        inp_residual = query if residual is None else residual
        if query_pos is not None:
            query = query + query_pos
        bs, nq, _ = query.shape
        plan = camera_plan if camera_plan is not None else CameraPlan(bev_mask)
        D = reference_points_cam.size(3)
        num_cams, l, _, _ = key.shape
        cams = torch.arange(self.num_cams, device=query.device)[:, None]
        valid = plan.valid[None, :, :, None]
        q_rebatch = query[:, plan.index] * valid                                               # [bs, cams, max_len, C]
        ref_rebatch = reference_points_cam.permute(1, 0, 2, 3, 4)[:, cams, plan.index] * valid[..., None]
        key = key.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        value = value.permute(2, 0, 1, 3).reshape(bs * self.num_cams, l, self.embed_dims)
        if plan.max_len == 0:                              # no camera sees any query (degenerate calibration)
            return self.dropout(self.output_proj(torch.zeros_like(query))) + inp_residual
        sampled = self.deformable_attention(query=q_rebatch.reshape(bs * self.num_cams, plan.max_len, self.embed_dims), key=key,
                                            value=value, reference_points=ref_rebatch.reshape(bs * self.num_cams, plan.max_len, D, 2),
                                            spatial_shapes=spatial_shapes, level_start_index=level_start_index)
        sampled = sampled.view(bs, self.num_cams, plan.max_len, self.embed_dims)
        slots = torch.zeros_like(query)
        for i, n in enumerate(plan.lens):                  # one camera at a time: unique rows, so the adds are ordered
            if n:
                slots.index_add_(1, plan.index[i, :n], sampled[:, i, :n])
        slots = self.output_proj(slots / plan.count[..., None])
        return self.dropout(slots) + inp_residual


@MODELS.register_module()
class TemporalSelfAttention(_DeformableAttention):
    """temporal_self_attention.py:25-264: deformable self attention over [previous BEV, current BEV] (num_bev_queue = 2)"""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, num_bev_queue=2, im2col_step=64, dropout=0.1,
                 batch_first=True, norm_cfg=None, init_cfg=None):
        super().__init__()
        self.norm_cfg, self.batch_first, self.im2col_step, self.num_bev_queue = norm_cfg, batch_first, im2col_step, num_bev_queue
        self.dropout = nn.Dropout(dropout)
        self._build(embed_dims, num_heads, num_levels, num_points, queue=num_bev_queue)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None,
                reference_points=None, spatial_shapes=None, level_start_index=None, flag="decoder", **kwargs):
        if value is None:
            assert self.batch_first
            bs, len_bev, c = query.shape
            value = torch.stack([query, query], 1).reshape(bs * 2, len_bev, c)
        identity = query if identity is None else identity
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, nq, dims = query.shape
        _, nv, _ = value.shape
        assert _num_keys(spatial_shapes) == nv
        assert self.num_bev_queue == 2
        NQ, NH, NL, NP = self.num_bev_queue, self.num_heads, self.num_levels, self.num_points
        query = torch.cat([value[:bs], query], -1)
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.reshape(bs * NQ, nv, NH, -1)
        offsets = self.sampling_offsets(query).view(bs, nq, NH, NQ, NL, NP, 2)
        weights = self.attention_weights(query).view(bs, nq, NH, NQ, NL * NP).softmax(-1).view(bs, nq, NH, NQ, NL, NP)
        weights = weights.permute(0, 3, 1, 2, 4, 5).reshape(bs * NQ, nq, NH, NL, NP).contiguous()
        offsets = offsets.permute(0, 3, 1, 2, 4, 5, 6).reshape(bs * NQ, nq, NH, NL, NP, 2)
        loc = self._locations(reference_points, offsets, spatial_shapes, NP)
        out = multi_scale_deformable_attn(value, spatial_shapes, level_start_index, loc, weights, self.im2col_step)
        out = out.view(bs, NQ, nq, dims).mean(1)            # == permute(1,2,0).view(nq, dims, bs, NQ).mean(-1).permute(2,0,1)
        out = self.output_proj(out)
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + identity


# ---- BEVFormer encoder ---------------------------------------------------------------------------------------------------
@MODELS.register_module()
class BEVFormerLayer(MyCustomBaseTransformerLayer):
    """encoder.py:232-400: temporal self attention -> norm -> spatial cross attention -> norm -> ffn -> norm"""

    def __init__(self, attn_cfgs, feedforward_channels, ffn_dropout=0.0, operation_order=None,
                 act_cfg=dict(type="ReLU", inplace=True), norm_cfg=dict(type="LN"), ffn_num_fcs=2, **kwargs):
        super().__init__(attn_cfgs=attn_cfgs, feedforward_channels=feedforward_channels, ffn_dropout=ffn_dropout,
                         operation_order=operation_order, norm_cfg=norm_cfg, ffn_num_fcs=ffn_num_fcs, **kwargs)
        assert len(operation_order) == 6 and set(operation_order) == {"self_attn", "norm", "cross_attn", "ffn"}

    def forward(self, query, key=None, value=None, bev_pos=None, query_pos=None, key_pos=None, attn_masks=None,
                query_key_padding_mask=None, key_padding_mask=None, ref_2d=None, ref_3d=None, bev_h=None, bev_w=None,
                reference_points_cam=None, mask=None, spatial_shapes=None, level_start_index=None, prev_bev=None, **kwargs):
        attn_masks = self._masks(attn_masks)
        ni = ai = fi = 0
        identity = query
        bev_level = level_tensors([(bev_h, bev_w)], query.device)
        for op in self.operation_order:
            if op == "self_attn":
                query = self.attentions[ai](query, prev_bev, prev_bev, identity if self.pre_norm else None, query_pos=bev_pos,
                                            key_pos=bev_pos, attn_mask=attn_masks[ai], key_padding_mask=query_key_padding_mask,
                                            reference_points=ref_2d, spatial_shapes=bev_level[0], level_start_index=bev_level[1],
                                            **kwargs)
                ai += 1
                identity = query
            elif op == "norm":
                query = self.norms[ni](query)
                ni += 1
            elif op == "cross_attn":
                query = self.attentions[ai](query, key, value, identity if self.pre_norm else None, query_pos=query_pos,
                                            key_pos=key_pos, reference_points=ref_3d, reference_points_cam=reference_points_cam,
                                            mask=mask, attn_mask=attn_masks[ai], key_padding_mask=key_padding_mask,
                                            spatial_shapes=spatial_shapes, level_start_index=level_start_index, **kwargs)
                ai += 1
                identity = query
            else:
                query = self.ffns[fi](query, identity if self.pre_norm else None)
                fi += 1
        return query


@MODELS.register_module()
class BEVFormerEncoder(TransformerLayerSequence):
    """encoder.py:29-229"""

    def __init__(self, *args, pc_range=None, num_points_in_pillar=4, return_intermediate=False, dataset_type="nuscenes", **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate, self.num_points_in_pillar, self.pc_range = return_intermediate, num_points_in_pillar, pc_range

    @staticmethod
    def get_reference_points(H, W, Z=8, num_points_in_pillar=4, dim="3d", bs=1, device="cuda", dtype=torch.float):
        """:50-88  '3d': [bs, D, H*W, 3] pillar points in [0, 1]^3; '2d': [bs, H*W, 1, 2] cell centres in [0, 1]^2"""
        lin = lambda n, steps: torch.linspace(0.5, n - 0.5, steps, dtype=dtype, device=device)
        if dim == "3d":
            D = num_points_in_pillar
            zs = lin(Z, D).view(D, 1, 1).expand(D, H, W) / Z
            xs = lin(W, W).view(1, 1, W).expand(D, H, W) / W
            ys = lin(H, H).view(1, H, 1).expand(D, H, W) / H
            ref = torch.stack((xs, ys, zs), -1).reshape(D, H * W, 3)
            return ref[None].repeat(bs, 1, 1, 1)
        ry, rx = torch.meshgrid(lin(H, H), lin(W, W), indexing="ij")
        ref = torch.stack((rx.reshape(-1)[None] / W, ry.reshape(-1)[None] / H), -1)
        return ref.repeat(bs, 1, 1).unsqueeze(2)

    def point_sampling(self, reference_points, pc_range, img_metas):
        """:92-152 -> reference_points_cam [num_cam, B, Q, D, 2] (image-normalised), bev_mask [num_cam, B, Q, D]"""
        lidar2img = L.h2d_like(reference_points, np.asarray([m["lidar2img"] for m in img_metas], dtype=np.float64)).float()
        B, D, Q, _ = reference_points.shape
        lo = L.h2d_like(reference_points, [float(v) for v in pc_range[:3]])
        ext = L.h2d_like(reference_points, [pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2]])
        pts = reference_points.float() * ext + lo
        pts = torch.cat((pts, torch.ones_like(pts[..., :1])), -1).reshape(B, 1, D * Q, 4)       # homogeneous
        cam = torch.matmul(pts, lidar2img.transpose(-1, -2)).view(B, -1, D, Q, 4)              # [B, N, D, Q, 4]
        eps = 1e-5
        depth = cam[..., 2:3]
        mask = depth > eps
        xy = cam[..., 0:2] / torch.maximum(depth, torch.ones_like(depth) * eps)
        H, W = img_metas[0]["img_shape"][0][0], img_metas[0]["img_shape"][0][1]
        xy = xy / L.h2d_like(xy, [W, H])
        mask = mask & (xy[..., 1:2] > 0.0) & (xy[..., 1:2] < 1.0) & (xy[..., 0:1] < 1.0) & (xy[..., 0:1] > 0.0)
        return xy.permute(1, 0, 3, 2, 4), mask.permute(1, 0, 3, 2, 4).squeeze(-1)

    def forward(self, bev_query, key, value, *args, bev_h=None, bev_w=None, bev_pos=None, spatial_shapes=None,
                level_start_index=None, valid_ratios=None, prev_bev=None, shift=0.0, **kwargs):
        bs = bev_query.size(1)
        ref_3d = self.get_reference_points(bev_h, bev_w, self.pc_range[5] - self.pc_range[2], self.num_points_in_pillar, "3d",
                                           bs, bev_query.device, bev_query.dtype)
        ref_2d = self.get_reference_points(bev_h, bev_w, dim="2d", bs=bs, device=bev_query.device, dtype=bev_query.dtype)
        reference_points_cam, bev_mask = self.point_sampling(ref_3d, self.pc_range, kwargs["img_metas"])
        # encoder.py:199-200 adds the shift in place, so BOTH halves of the hybrid reference carry it (kept)
        ref_2d = ref_2d + shift[:, None, None, :]
        bev_query, bev_pos = bev_query.permute(1, 0, 2), bev_pos.permute(1, 0, 2)
        len_bev, nlvl = ref_2d.shape[1], ref_2d.shape[2]
        if prev_bev is not None:
            prev_bev = torch.stack([prev_bev.permute(1, 0, 2), bev_query], 1).reshape(bs * 2, len_bev, -1)
        hybrid_ref_2d = torch.stack([ref_2d, ref_2d], 1).reshape(bs * 2, len_bev, nlvl, 2)
        plan = CameraPlan(bev_mask)
        intermediate = []
        for layer in self.layers:
            bev_query = layer(bev_query, key, value, *args, bev_pos=bev_pos, ref_2d=hybrid_ref_2d, ref_3d=ref_3d, bev_h=bev_h,
                              bev_w=bev_w, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                              reference_points_cam=reference_points_cam, bev_mask=bev_mask, prev_bev=prev_bev,
                              camera_plan=plan, **kwargs)
            if self.return_intermediate:
                intermediate.append(bev_query)
        return torch.stack(intermediate) if self.return_intermediate else bev_query


# ---- decoder -------------------------------------------------------------------------------------------------------------
@MODELS.register_module()
class DetectionTransformerDecoder(TransformerLayerSequence):
    """decoder.py:46-128: per layer, refine the 3-D reference points with that layer's regression branch."""

    def __init__(self, *args, return_intermediate=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.return_intermediate = return_intermediate

    def forward(self, query, *args, reference_points=None, reg_branches=None, key_padding_mask=None, **kwargs):
        output, inter, inter_ref = query, [], []
        for lid, layer in enumerate(self.layers):
            output = layer(output, *args, reference_points=reference_points[..., :2].unsqueeze(2),
                           key_padding_mask=key_padding_mask, **kwargs)
            if reg_branches is not None:
                tmp = reg_branches[lid](output.permute(1, 0, 2))
                assert reference_points.shape[-1] == 3
                new_ref = torch.cat((tmp[..., :2] + inverse_sigmoid(reference_points[..., :2]),
                                     tmp[..., 4:5] + inverse_sigmoid(reference_points[..., 2:3])), -1).sigmoid()
                reference_points = new_ref.detach()
            if self.return_intermediate:
                inter.append(output)
                inter_ref.append(reference_points)
        if self.return_intermediate:
            return torch.stack(inter), torch.stack(inter_ref)
        return output, reference_points


# ---- whole transformers ----------------------------------------------------------------------------------------------------
def rotate_nearest(img, angle, center):
    """torchvision.transforms.functional.rotate(img [C, H, W], angle (degrees, counter-clockwise), center=(x, y)) with its
    defaults (nearest interpolation, no expansion, zero fill) -- torchvision is un-vendored (perception_transformer.py:10,141):
    inverse affine map about ``center`` on pixel-centre coordinates, sampled with grid_sample(align_corners=False)."""
    C, H, W = img.shape
    cx, cy = center[0] - W * 0.5, center[1] - H * 0.5
    rot = math.radians(-angle)
    m = [math.cos(rot), math.sin(rot), 0.0, -math.sin(rot), math.cos(rot), 0.0]
    m[2] += m[0] * (-cx) + m[1] * (-cy) + cx
    m[5] += m[3] * (-cx) + m[4] * (-cy) + cy
    theta = L.h2d_like(img, m).reshape(1, 2, 3)
    base = torch.empty(1, H, W, 3, dtype=img.dtype, device=img.device)
    base[..., 0] = torch.linspace(-W * 0.5 + 0.5, W * 0.5 + 0.5 - 1, steps=W, device=img.device, dtype=img.dtype)
    base[..., 1] = torch.linspace(-H * 0.5 + 0.5, H * 0.5 + 0.5 - 1, steps=H, device=img.device, dtype=img.dtype).unsqueeze(-1)
    base[..., 2] = 1
    rescaled = theta.transpose(1, 2) / L.h2d_like(img, [0.5 * W, 0.5 * H])
    grid = base.view(1, H * W, 3).bmm(rescaled).view(1, H, W, 2)
    return torch.nn.functional.grid_sample(img[None], grid, mode="nearest", padding_mode="zeros", align_corners=False)[0]


@MODELS.register_module()
class PerceptionTransformer(nn.Module):
    """perception_transformer.py:22-280"""

    def __init__(self, num_feature_levels=4, num_cams=6, two_stage_num_proposals=300, encoder=None, decoder=None, embed_dims=256,
                 rotate_prev_bev=True, use_shift=True, use_can_bus=True, can_bus_norm=True, use_cams_embeds=True,
                 rotate_center=[100, 100], **kwargs):
        super().__init__()
        self.encoder = build_transformer_layer_sequence(encoder)
        self.decoder = build_transformer_layer_sequence(decoder)
        self.embed_dims, self.num_feature_levels, self.num_cams = embed_dims, num_feature_levels, num_cams
        self.rotate_prev_bev, self.use_shift, self.use_can_bus = rotate_prev_bev, use_shift, use_can_bus
        self.can_bus_norm, self.use_cams_embeds = can_bus_norm, use_cams_embeds
        self.two_stage_num_proposals, self.rotate_center = two_stage_num_proposals, rotate_center
        self.level_embeds = nn.Parameter(torch.Tensor(num_feature_levels, embed_dims))
        self.cams_embeds = nn.Parameter(torch.Tensor(num_cams, embed_dims))
        self.reference_points = nn.Linear(embed_dims, 3)
        self.can_bus_mlp = nn.Sequential(nn.Linear(18, embed_dims // 2), nn.ReLU(inplace=True),
                                         nn.Linear(embed_dims // 2, embed_dims), nn.ReLU(inplace=True))
        if can_bus_norm:
            self.can_bus_mlp.add_module("norm", nn.LayerNorm(embed_dims))

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (MSDeformableAttention3D, TemporalSelfAttention, CustomMSDeformableAttention)):
                m.init_weights()
        nn.init.normal_(self.level_embeds)
        nn.init.normal_(self.cams_embeds)
        xavier_init(self.reference_points, distribution="uniform", bias=0.0)

    def get_bev_features(self, mlvl_feats, bev_queries, bev_h, bev_w, grid_length=[0.512, 0.512], bev_pos=None, prev_bev=None,
                         **kwargs):
        """:103-208 -> bev_embed [bs, bev_h * bev_w, embed_dims]"""
        metas = kwargs["img_metas"]
        bs = mlvl_feats[0].size(0)
        bev_queries = bev_queries.unsqueeze(1).repeat(1, bs, 1)
        bev_pos = bev_pos.flatten(2).permute(2, 0, 1)
        # ego motion -> BEV shift (host floats from the can bus, as the reference)
        dx = np.array([m["can_bus"][0] for m in metas])
        dy = np.array([m["can_bus"][1] for m in metas])
        ego_angle = np.array([m["can_bus"][-2] / np.pi * 180 for m in metas])
        length = np.sqrt(dx ** 2 + dy ** 2)
        bev_angle = ego_angle - np.arctan2(dy, dx) / np.pi * 180
        shift_y = length * np.cos(bev_angle / 180 * np.pi) / grid_length[0] / bev_h * self.use_shift
        shift_x = length * np.sin(bev_angle / 180 * np.pi) / grid_length[1] / bev_w * self.use_shift
        shift = L.h2d_like(bev_queries, np.array([shift_x, shift_y])).permute(1, 0)
        if prev_bev is not None:
            if prev_bev.shape[1] == bev_h * bev_w:
                prev_bev = prev_bev.permute(1, 0, 2)
            if self.rotate_prev_bev:
                prev_bev = prev_bev.clone()
                for i in range(bs):
                    tmp = prev_bev[:, i].reshape(bev_h, bev_w, -1).permute(2, 0, 1)
                    tmp = rotate_nearest(tmp, metas[i]["can_bus"][-1], center=self.rotate_center)
                    prev_bev[:, i] = tmp.permute(1, 2, 0).reshape(bev_h * bev_w, -1)
        can_bus = L.h2d_like(bev_queries, np.array([m["can_bus"] for m in metas]))
        bev_queries = bev_queries + self.can_bus_mlp(can_bus)[None, :, :] * self.use_can_bus
        feat_flatten, spatial_shapes = [], []
        for lvl, feat in enumerate(mlvl_feats):
            bs, num_cam, c, h, w = feat.shape
            feat = feat.flatten(3).permute(1, 0, 3, 2)
            if self.use_cams_embeds:
                feat = feat + self.cams_embeds[:, None, None, :].to(feat.dtype)
            feat = feat + self.level_embeds[None, None, lvl:lvl + 1, :].to(feat.dtype)
            spatial_shapes.append((h, w))
            feat_flatten.append(feat)
        feat_flatten = torch.cat(feat_flatten, 2).permute(0, 2, 1, 3)           # [num_cam, sum(H*W), bs, C]
        spatial_shapes, level_start_index = level_tensors(spatial_shapes, bev_pos.device)
        return self.encoder(bev_queries, feat_flatten, feat_flatten, bev_h=bev_h, bev_w=bev_w, bev_pos=bev_pos,
                            spatial_shapes=spatial_shapes, level_start_index=level_start_index, prev_bev=prev_bev,
                            shift=shift, **kwargs)

    def forward(self, mlvl_feats, bev_queries, object_query_embed, bev_h, bev_w, grid_length=[0.512, 0.512], bev_pos=None,
                reg_branches=None, cls_branches=None, prev_bev=None, **kwargs):
        """:211-280 -> bev_embed [bev_h*bev_w, bs, C], decoder states [layers, num_query, bs, C], references"""
        bev_embed = self.get_bev_features(mlvl_feats, bev_queries, bev_h, bev_w, grid_length=grid_length, bev_pos=bev_pos,
                                          prev_bev=prev_bev, **kwargs)
        bs = mlvl_feats[0].size(0)
        query_pos, query = torch.split(object_query_embed, self.embed_dims, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        bev_embed = bev_embed.permute(1, 0, 2)
        bev_level = level_tensors([(bev_h, bev_w)], query.device)
        inter_states, inter_references = self.decoder(
            query=query.permute(1, 0, 2), key=None, value=bev_embed, query_pos=query_pos.permute(1, 0, 2),
            reference_points=reference_points, reg_branches=reg_branches, cls_branches=cls_branches,
            spatial_shapes=bev_level[0], level_start_index=bev_level[1], **kwargs)
        return bev_embed, inter_states, reference_points, inter_references


@MODELS.register_module()
class DeformableDetrTransformerDistill(nn.Module):
    """detr_transformer.py:24-355 (the teacher's head transformer): BEV queries attend the 4-level LiDAR feature pyramid
    through 6 deformable encoder layers -> bev_embed; the shared DETR decoder reads it.  ``as_two_stage`` is not configured
    by the reference's teachers (:272-291) and not built."""

    def __init__(self, as_two_stage=False, num_feature_levels=4, two_stage_num_proposals=300, bev_size=128, encoder=None,
                 decoder=None, init_cfg=None, **kwargs):
        super().__init__()
        assert not as_two_stage, "the two-stage variant is not configured by the reference's teacher configs"
        self.encoder = build_transformer_layer_sequence(encoder)
        self.decoder = build_transformer_layer_sequence(decoder)
        self.embed_dims = self.encoder.embed_dims
        self.as_two_stage, self.num_feature_levels, self.two_stage_num_proposals = as_two_stage, num_feature_levels, two_stage_num_proposals
        self.bev_size = bev_size
        self.level_embeds = nn.Parameter(torch.Tensor(num_feature_levels, self.embed_dims))
        self.reference_points = nn.Linear(self.embed_dims, 3)

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if type(m) is MultiScaleDeformableAttention:
                m.init_weights()
        xavier_init(self.reference_points, distribution="uniform", bias=0.0)
        nn.init.normal_(self.level_embeds)

    @staticmethod
    def get_reference_points(bs, spatial_shapes, bev_size, device):
        """:127-150 -> [bs, bev_size**2, 1, 2] cell centres"""
        lin = torch.linspace(0.5, bev_size - 0.5, bev_size, dtype=torch.float32, device=device)
        ry, rx = torch.meshgrid(lin, lin, indexing="ij")
        ref = torch.stack((rx.reshape(-1)[None] / bev_size, ry.reshape(-1)[None] / bev_size), -1)
        return ref.repeat(bs, 1, 1).unsqueeze(2)

    def forward(self, mlvl_feats, bev_queries, mlvl_masks, bev_mask, bev_pos, query_embed, mlvl_pos_embeds, reg_branches=None,
                cls_branches=None, only_bev=False, **kwargs):
        """:181-355 -> (decoder states, init reference, inter references, bev_embed [bs, bev_size**2, C], None, None)"""
        assert query_embed is not None
        feat_flatten, spatial_shapes = [], []
        for feat in mlvl_feats:
            bs, c, h, w = feat.shape
            spatial_shapes.append((h, w))
            feat_flatten.append(feat.flatten(2).transpose(1, 2))
        feat_flatten = torch.cat(feat_flatten, 1)
        spatial_shapes, level_start_index = level_tensors(spatial_shapes, feat_flatten.device)
        reference_points = self.get_reference_points(bs, spatial_shapes, self.bev_size, device=feat_flatten.device)
        feat_flatten = feat_flatten.permute(1, 0, 2)
        memory = self.encoder(query=bev_queries.unsqueeze(1).repeat(1, bs, 1), key=feat_flatten, value=feat_flatten,
                              query_pos=bev_pos.flatten(2).permute(2, 0, 1), query_key_padding_mask=bev_mask.flatten(1),
                              spatial_shapes=spatial_shapes, reference_points=reference_points,
                              level_start_index=level_start_index, **kwargs)
        bev_embed = memory.permute(1, 0, 2)
        if only_bev:                   # the distillation reads the BEV embedding only: the decoder is skipped
            return None, None, None, bev_embed, None, None
        c = bev_embed.shape[-1]
        query_pos, query = torch.split(query_embed, c, dim=1)
        query_pos = query_pos.unsqueeze(0).expand(bs, -1, -1)
        query = query.unsqueeze(0).expand(bs, -1, -1)
        reference_points = self.reference_points(query_pos).sigmoid()
        bev_level = level_tensors([(self.bev_size, self.bev_size)], query.device)
        inter_states, inter_references = self.decoder(
            query=query.permute(1, 0, 2), key=None, value=memory, query_pos=query_pos.permute(1, 0, 2),
            reference_points=reference_points, reg_branches=reg_branches, cls_branches=cls_branches,
            spatial_shapes=bev_level[0], level_start_index=bev_level[1], **kwargs)
        return inter_states, reference_points, inter_references, bev_embed, None, None
