"""Oracle (CPU, torch) for multi-scale deformable attention -- TEST INFRASTRUCTURE ONLY.

The arithmetic lives in mmcv-full==1.6.0 (`mmcv/ops/csrc/.../ms_deform_attn_cuda_kernel.cuh`, `mmcv/ops/
multi_scale_deform_attn.py: multi_scale_deformable_attn_pytorch`), which is NOT under /root/reference (un-vendored
dependency pinned in docker/Dockerfile:21; call site mmdet3d/models/transformer_modules/
multi_scale_deformable_attn_function.py:10-12).  PARITY UNPINNED by reference outputs: this file restates the published
definition two independent ways that are checked against each other (tests/test_oracle_msda.py):
  * `msda_grid_sample`  the grid_sample formulation (the op's documented pure-PyTorch equivalent): per level,
    F.grid_sample(value_l, 2 * loc - 1, mode='bilinear', padding_mode='zeros', align_corners=False), weighted sum;
  * `msda_naive`        explicit python loops over (b, q, h, l, p) with the CUDA kernel's corner rule
    (h_im = y * H - 0.5, every corner outside the map contributes 0).
"""
import math

import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B, S, NH, D], spatial_shapes [(H, W)], sampling_locations [B, Q, NH, L, P, 2], attention_weights
    [B, Q, NH, L, P] -> [B, Q, NH * D]; differentiable (autograd = the backward the kernels are held to)."""
    B, S, NH, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    vals = value.split([h * w for h, w in spatial_shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for l, (h, w) in enumerate(spatial_shapes):
        v = vals[l].flatten(2).transpose(1, 2).reshape(B * NH, D, h, w)
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)                  # [B*NH, Q, P, 2]
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    att = attention_weights.transpose(1, 2).reshape(B * NH, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * att).sum(-1).view(B, NH * D, Q)
    return out.transpose(1, 2).contiguous()


def msda_naive(value, spatial_shapes, sampling_locations, attention_weights):
    B, S, NH, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    out = torch.zeros((B, Q, NH, D), dtype=value.dtype)
    starts = [0]
    for h, w in spatial_shapes:
        starts.append(starts[-1] + h * w)
    for b in range(B):
        for q in range(Q):
            for hd in range(NH):
                for l, (H, W) in enumerate(spatial_shapes):
                    for p in range(P):
                        x, y = sampling_locations[b, q, hd, l, p].tolist()
                        h_im, w_im = y * H - 0.5, x * W - 0.5
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                            continue
                        h0, w0 = math.floor(h_im), math.floor(w_im)
                        lh, lw = h_im - h0, w_im - w0
                        acc = torch.zeros(D, dtype=value.dtype)
                        for (yy, xx, c) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                            (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                            if 0 <= yy <= H - 1 and 0 <= xx <= W - 1:
                                acc = acc + c * value[b, starts[l] + yy * W + xx, hd]
                        out[b, q, hd] += attention_weights[b, q, hd, l, p] * acc
    return out.view(B, Q, NH * D)
