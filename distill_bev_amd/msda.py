"""Multi-scale deformable attention -- Python mirror of the reference's op surface
(``mmdet3d/models/transformer_modules/multi_scale_deformable_attn_function.py``:
``MultiScaleDeformableAttnFunction_fp32`` :89-162, ``MultiScaleDeformableAttnFunction_fp16`` :15-86) on the gfx950 kernels
``dbev_msda_forward / dbev_msda_backward`` (csrc/msda.hip) instead of mmcv's ``_ext.ms_deform_attn_*``.

Same call: ``Function.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
attention_weights, im2col_step)`` -> ``[bs, num_queries, embed_dims]``; gradients for value, sampling_locations and
attention_weights (``im2col_step`` only tiled mmcv's launch over the batch: the kernels here take the whole batch in
one launch and ignore it).  The backward's value gradient is a deterministic gather (mmcv: float atomics).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib as L


_LEVELS = {}


def level_tensors(shapes, device):
    """(spatial_shapes [L, 2] int64, level_start_index [L] int64) on `device` for the host list ``shapes`` = [(h, w), ...],
    built once per geometry and device.  The tensors carry their host values, so the op never reads them back: the
    reference's modules rebuild both tensors in every layer call and mmcv's op reads them on the device."""
    key = (tuple((int(h), int(w)) for h, w in shapes), str(device))
    hit = _LEVELS.get(key)
    if hit is None:
        hw = [v for pair in key[0] for v in pair]
        st, acc = [], 0
        for h, w in key[0]:
            st.append(acc)
            acc += h * w
        ss = L.h2d(torch.tensor(key[0], dtype=torch.long), device)
        ls = L.h2d(torch.tensor(st, dtype=torch.long), device)
        ss._dbev_host, ls._dbev_host = hw, st
        hit = _LEVELS[key] = (ss, ls)
    return hit


def _levels(spatial_shapes, level_start_index):
    hw = getattr(spatial_shapes, "_dbev_host", None)
    st = getattr(level_start_index, "_dbev_host", None)
    if hw is None:
        hw = [int(v) for v in spatial_shapes.reshape(-1).tolist()]       # one small device read-back per call
    if st is None:
        st = [int(v) for v in level_start_index.reshape(-1).tolist()]
    return hw, st


class MultiScaleDeformableAttnFunction_fp32(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=64):
        dev = L.require_cuda(value, sampling_locations, attention_weights)
        value = value.float().contiguous()
        loc = sampling_locations.float().contiguous()
        att = attention_weights.float().contiguous()
        B, S, NH, D = value.shape
        _, Q, _, Lv, P, _ = loc.shape
        assert att.shape == (B, Q, NH, Lv, P) and loc.shape[2] == NH
        hw, st = _levels(value_spatial_shapes, value_level_start_index)      # a few host ints (one small D2H per call)
        out = torch.empty((B, Q, NH * D), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_msda_forward", L.ptr(value), L.host_ints(hw), L.host_ints(st), L.ptr(loc), L.ptr(att), B, S, NH, D,
                   Q, Lv, P, L.ptr(out), L.stream_ptr(dev))
        ctx.save_for_backward(value, loc, att)
        ctx.levels = (hw, st)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, loc, att = ctx.saved_tensors
        hw, st = ctx.levels
        dev = value.device
        B, S, NH, D = value.shape
        _, Q, _, Lv, P, _ = loc.shape
        go = grad_output.float().contiguous()
        gv = torch.empty_like(value)
        gl = torch.empty_like(loc)
        ga = torch.empty_like(att)
        with torch.cuda.device(dev):
            nbytes = int(L.call("dbev_msda_backward_workspace_bytes", B, S, NH, Q, Lv, P))
            if nbytes == 0:
                raise L.DbevHipError("multi-scale deformable attention backward: batch too large for 32-bit sample ids / bin offsets")
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            L.call("dbev_msda_backward", L.ptr(value), L.host_ints(hw), L.host_ints(st), L.ptr(loc), L.ptr(att), L.ptr(go),
                   B, S, NH, D, Q, Lv, P, L.ptr(gv), L.ptr(gl), L.ptr(ga), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
        return gv, None, None, gl, ga, None


class MultiScaleDeformableAttnFunction_fp16(Function):
    """:15-86 casts the inputs to half and runs mmcv's half kernel.  Here the SAME fp32 kernels run on the up-cast
    operands (fp32 accumulation, half in / half out): at least the reference's precision."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step=64):
        ctx.dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        with torch.enable_grad():
            v = value.detach().float().requires_grad_(True)
            l = sampling_locations.detach().float().requires_grad_(True)
            a = attention_weights.detach().float().requires_grad_(True)
            out = MultiScaleDeformableAttnFunction_fp32.apply(v, value_spatial_shapes, value_level_start_index, l, a,
                                                              im2col_step)
        ctx.inner = (v, l, a, out)
        return out.detach().half()

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        v, l, a, out = ctx.inner
        gv, gl, ga = torch.autograd.grad(out, (v, l, a), grad_output.float())
        dv, dl, da = ctx.dtypes
        return gv.to(dv), None, None, gl.to(dl), ga.to(da), None


def multi_scale_deformable_attn(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                attention_weights, im2col_step=64):
    return MultiScaleDeformableAttnFunction_fp32.apply(value, value_spatial_shapes, value_level_start_index,
                                                       sampling_locations, attention_weights, im2col_step)
