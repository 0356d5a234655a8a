"""Sparse convolution -- Python mirror of the reference's ``mmdet3d.ops.spconv`` surface (spconv v1.x fork:
``structure.py`` SparseConvTensor :21-69, ``modules.py`` SparseModule / SparseSequential / ToDense :43-196, ``conv.py``
SparseConvolution :60-225 and its SparseConv{2,3}d / SubMConv{2,3}d / SparseInverseConv{2,3}d subclasses,
``ops.py`` get_conv_output_size / get_indice_pairs / indice_conv :19-126, ``functional.py``) on the gfx950 kernels of
csrc/spconv.hip instead of the CUDA-only ``sparse_conv_ext``.

Differences that are MI355X design choices, not semantics:
  * the rulebook is output-stationary (``Rulebook.nbr [n_out, K]``): one gather-GEMM kernel per layer instead of K
    gather / GEMM / scatter-add rounds; the reference's pair lists (``indice_pairs [K, 2, N]``, ``indice_pair_num [K]``)
    are still produced and cached under ``SparseConvTensor.indice_dict[indice_key]`` in the reference's tuple layout;
  * rulebooks are also cached for layers WITHOUT an ``indice_key`` (the reference rebuilds them: every SubMConv3d of a
    SparseBasicBlock, sparse_block.py:66-121, recomputes the same pairs), keyed on the index tensor and the geometry;
  * the backward (``indice_conv_backward``) runs per kernel offset on torch GEMMs over the pair lists -- the voxel teachers
    are frozen (``torch.no_grad``) on the distillation path, only the forward is hot.
"""
import math
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn import init
from torch.nn.parameter import Parameter

from . import _lib as L
from .registry import register_conv


# ---- ops.py ------------------------------------------------------------------------------------------------------------
def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """ops.py:19-30"""
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """ops.py:33-43"""
    out = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out


def _t3(v, ndim):
    v = list(v) if isinstance(v, (list, tuple)) else [v] * ndim
    return [1] * (3 - len(v)) + [int(x) for x in v] if len(v) < 3 else [int(x) for x in v]


class Rulebook:
    """One convolution geometry on one index set: the neighbour table the kernel reads, plus -- on demand -- the reference's
    pair lists (the backward pass and get_indice_pairs() read them) and the inverse table (SparseInverseConv3d and the data
    gradient read it); an inference forward of the encoder needs neither."""
    __slots__ = ("nbr", "_inv", "outids", "_pairs", "_pair_num", "n_in", "n_out", "K", "out_shape", "_pn_host")

    @property
    def inv(self):
        if self._inv is None:
            dev = self.nbr.device
            self._inv = torch.empty((max(self.n_in, 1), self.K), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                L.call("dbev_spconv_inverse_table", L.ptr(self.nbr), self.n_out, self.K, self.n_in, L.ptr(self._inv),
                       L.stream_ptr(dev))
        return self._inv

    def _build_pairs(self):
        dev = self.nbr.device
        self._pairs = torch.empty((self.K, 2, max(self.n_in, 1)), dtype=torch.int32, device=dev)
        self._pair_num = torch.zeros((self.K,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            nbytes = int(L.call("dbev_spconv_pair_lists_workspace_bytes", self.n_out, self.K))
            ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
            L.call("dbev_spconv_pair_lists", L.ptr(self.nbr), self.n_out, self.K, self.n_in, L.ptr(self._pairs),
                   L.ptr(self._pair_num), L.ptr(ws), ws.numel(), L.stream_ptr(dev))

    @property
    def indice_pairs(self):
        if self._pairs is None:
            self._build_pairs()
        return self._pairs

    @property
    def indice_pair_num(self):
        if self._pair_num is None:
            self._build_pairs()
        return self._pair_num

    def pair_num_host(self):
        if self._pn_host is None:
            self._pn_host = self.indice_pair_num.cpu().tolist()
        return self._pn_host


class IndiceData(object):
    """The reference's cache entry (outids, indices, indice_pairs, indice_pair_num, spatial_shape) of conv.py:176-180; reads
    like that 5-tuple, the pair lists materialise on first access."""

    def __init__(self, rulebook, indices, spatial_shape):
        self.rulebook, self.indices, self.spatial_shape = rulebook, indices, spatial_shape

    def _get(self, i):
        rb = self.rulebook
        return (lambda: rb.outids, lambda: self.indices, lambda: rb.indice_pairs, lambda: rb.indice_pair_num,
                lambda: self.spatial_shape)[i]()

    def __len__(self):
        return 5

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self._get(j) for j in range(5)[i])
        return self._get(i)

    def __iter__(self):
        return (self._get(j) for j in range(5))


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm, transposed=False, output_padding=0):
    """-> Rulebook.  indices int32 [n, ndim + 1] (batch first), ndim in (2, 3).  transposed: SparseConvTranspose* (an input at i
    reaches the outputs i * stride - padding + k * dilation, ops.py:73-76)."""
    dev = L.require_cuda(indices)
    ndim = indices.shape[1] - 1
    assert ndim in (2, 3), "sparse convolutions over 2 or 3 spatial dimensions"
    ks, st, pd, dl = _t3(ksize, ndim), _t3(stride, ndim), _t3(padding, ndim), _t3(dilation, ndim)
    if subm:      # spconv_ops.h:76-80: a submanifold convolution ignores the configured stride / padding (1, ksize // 2)
        st, pd = [1, 1, 1], [k // 2 for k in ks]
    in_shape = [1] * (3 - ndim) + [int(v) for v in spatial_shape]
    idx4 = indices.int().contiguous()
    if ndim == 2:
        idx4 = torch.cat([idx4[:, :1], torch.zeros_like(idx4[:, :1]), idx4[:, 1:]], dim=1).contiguous()
    assert not (subm and transposed)
    op = _t3(output_padding, ndim)
    if ndim == 2:                                 # the dummy z axis of a 2-D geometry: kernel 1, stride 1, no padding
        pd[0], op[0] = 0, 0
    if subm:
        out_shape = in_shape
    elif transposed:
        out_shape = get_deconv_output_size(in_shape, ks, st, pd, dl, op)
    else:
        out_shape = get_conv_output_size(in_shape, ks, st, pd, dl)
    tr = 1 if transposed else 0
    n_in = idx4.shape[0]
    K = ks[0] * ks[1] * ks[2]
    rb = Rulebook()
    rb.n_in, rb.K, rb._pn_host, rb._pairs, rb._pair_num, rb._inv = n_in, K, None, None, None, None
    rb.out_shape = out_shape[3 - ndim:]
    hi = L.host_ints
    with torch.cuda.device(dev):
        vol = batch_size * out_shape[0] * out_shape[1] * out_shape[2]
        max_out = n_in if subm else int(min(max(n_in * K, 1), vol))
        nbytes = int(L.call("dbev_spconv_build_workspace_bytes", n_in, batch_size, hi(in_shape), hi(out_shape), K, max_out))
        ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
        if subm:
            out4, n_out = idx4, n_in
        else:
            out4 = torch.empty((max(max_out, 1), 4), dtype=torch.int32, device=dev)
            n_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
            L.call("dbev_spconv_outputs", L.ptr(idx4), n_in, batch_size, hi(in_shape), hi(out_shape), hi(ks), hi(st), hi(pd),
                   hi(dl), tr, L.ptr(out4), max_out, L.ptr(n_dev), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
            n_out = int(n_dev.item())                       # num_act_out of the reference (one read-back per rulebook)
            out4 = out4[:n_out]
        rb.n_out = n_out
        rb.nbr = torch.empty((max(n_out, 1), K), dtype=torch.int32, device=dev)
        L.call("dbev_spconv_neighbors", L.ptr(idx4), n_in, L.ptr(out4), n_out, batch_size, hi(in_shape), hi(out_shape), hi(ks),
               hi(st), hi(pd), hi(dl), tr, L.ptr(rb.nbr), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(ws), ws.numel(),
               L.stream_ptr(dev))
    rb.outids = indices if subm else (out4 if ndim == 3 else out4[:, [0, 2, 3]].contiguous())
    return rb


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None):
    """ops.py:46-104 -> (outids, indice_pairs [K, 2, N], indice_pair_num [K])."""
    rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm, transpose, out_padding)
    return rb.outids, rb.indice_pairs, rb.indice_pair_num


def _pad16(n):
    return (n + 15) // 16 * 16


class _PairsView(object):
    """pair lists handed in by the caller (indice_conv) with the two tables derived from them: the members the backward reads
    from a Rulebook"""

    def __init__(self, indice_pairs, indice_pair_num, nbr, inv):
        self.indice_pairs, self.indice_pair_num, self.nbr, self.inv = indice_pairs, indice_pair_num, nbr, inv


def _conv_forward(features, weight, table, n_out, scale=None, shift=None, residual=None, relu=False):
    """dbev_spconv_forward_fused on channel counts padded to multiples of 16.  The kernel tiles up to 256 input and 128 output
    channels (the reference's SparseEncoder stops at 128); wider layers run it per slice: output slices are independent, the
    input slices of an output slice are summed in slice order and the epilogue (scale / shift / residual / ReLU) is applied to the sum."""
    K, Cin, Cout = weight.shape
    if Cin > 256 or Cout > 128:
        cuts = lambda n, s_: [(a, min(a + s_, n)) for a in range(0, n, s_)]
        outs = []
        for o0, o1 in cuts(Cout, 128):
            acc = None
            for i0, i1 in cuts(Cin, 256):
                part = _conv_forward_tile(features[:, i0:i1], weight[:, i0:i1, o0:o1], table, n_out)
                acc = part if acc is None else acc + part
            outs.append(acc)
        out = torch.cat(outs, 1)
        if scale is not None:
            out = out * scale
        if shift is not None:
            out = out + shift
        if residual is not None:
            out = out + residual
        return torch.relu(out) if relu else out
    return _conv_forward_tile(features, weight, table, n_out, scale, shift, residual, relu)


def _conv_forward_tile(features, weight, table, n_out, scale=None, shift=None, residual=None, relu=False):
    dev = L.require_cuda(features, weight)
    K, Cin, Cout = weight.shape
    f, w = features.float(), weight.float()
    ci, co = _pad16(Cin), _pad16(Cout)
    pad = torch.nn.functional.pad
    if ci != Cin:
        f, w = pad(f, (0, ci - Cin)), pad(w, (0, 0, 0, ci - Cin))
    if co != Cout:
        w = pad(w, (0, co - Cout))
        scale = None if scale is None else pad(scale, (0, co - Cout))
        shift = None if shift is None else pad(shift, (0, co - Cout))
        residual = None if residual is None else pad(residual, (0, co - Cout))
    c = lambda t: None if t is None else t.float().contiguous()
    f, w, scale, shift, residual = c(f), c(w), c(scale), c(shift), c(residual)
    out = torch.empty((n_out, co), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.call("dbev_spconv_forward_fused", L.ptr(f), L.ptr(w), L.ptr(scale), L.ptr(shift), L.ptr(residual), int(bool(relu)),
               L.ptr(table), n_out, K, ci, co, L.ptr(out), L.stream_ptr(dev))
    return out[:, :Cout] if co != Cout else out


def _pad_pow2(n):
    """channel count the weight-gradient kernel tiles: 16, 32, 64 or 128"""
    c = 16
    while c < n:
        c *= 2
    return c


def _conv_backward(features, weight, grad_out, book, swap, need_input_grad, need_weight_grad):
    """indice_conv_backward (spconv_ops.h:352-420) on the kernels: the data gradient is the forward gather-GEMM over the table of
    the opposite direction with transposed weights, the weight gradient one MFMA GEMM per kernel offset over that offset's
    compacted pair list.  No host read-back, no float atomics.  The kernels tile up to 128 channels a side; wider layers
    (the reference's SparseEncoder stops at 128, but the module surface takes any width, as the forward does) run them per
    128-channel slice pair: the weight gradient of a slice pair is independent of the others, the data gradient of an input
    slice is the sum over the output slices (added in slice order: still bit-reproducible)."""
    Cin, Cout = weight.shape[1], weight.shape[2]
    if Cin <= 128 and Cout <= 128:
        return _conv_backward_tile(features, weight, grad_out, book, swap, need_input_grad, need_weight_grad)
    cuts = lambda n: [(a, min(a + 128, n)) for a in range(0, n, 128)]
    gin = torch.empty((features.shape[0], Cin), dtype=torch.float32, device=features.device) if need_input_grad else None
    gw = torch.empty((weight.shape[0], Cin, Cout), dtype=torch.float32, device=features.device) if need_weight_grad else None
    for i0, i1 in cuts(Cin):
        acc = None
        for o0, o1 in cuts(Cout):
            gi, gwp = _conv_backward_tile(features[:, i0:i1], weight[:, i0:i1, o0:o1], grad_out[:, o0:o1], book, swap,
                                          need_input_grad, need_weight_grad)
            if need_weight_grad:
                gw[:, i0:i1, o0:o1] = gwp
            if need_input_grad:
                acc = gi if acc is None else acc + gi
        if need_input_grad:
            gin[:, i0:i1] = acc
    return gin, gw


def _conv_backward_tile(features, weight, grad_out, book, swap, need_input_grad, need_weight_grad):
    """one call of the two backward kernels, <= 128 channels a side (padded to 16 / 32 / 64 / 128)"""
    dev = L.require_cuda(features, weight, grad_out)
    K, Cin, Cout = weight.shape
    ci, co = _pad_pow2(Cin), _pad_pow2(Cout)
    assert ci <= 128 and co <= 128
    pad = torch.nn.functional.pad
    f, g, w = features.float(), grad_out.float(), weight.float()
    if ci != Cin:
        f, w = pad(f, (0, ci - Cin)), pad(w, (0, 0, 0, ci - Cin))
    if co != Cout:
        g, w = pad(g, (0, co - Cout)), pad(w, (0, co - Cout))
    f, g, w = f.contiguous(), g.contiguous(), w.contiguous()
    n_in = f.shape[0]
    gin = gw = None
    with torch.cuda.device(dev):
        if need_input_grad:
            table = book.nbr if swap else book.inv          # rows of this layer's INPUT <- rows of its output, per offset
            gin = torch.empty((n_in, ci), dtype=torch.float32, device=dev)
            ws = torch.empty((K * ci * co,), dtype=torch.float32, device=dev)
            L.call("dbev_spconv_backward_data", L.ptr(g), L.ptr(w), L.ptr(table), n_in, K, ci, co, L.ptr(gin), L.ptr(ws),
                   ws.numel() * 4, L.stream_ptr(dev))
            gin = gin[:, :Cin] if ci != Cin else gin
        if need_weight_grad:
            pairs, nums = book.indice_pairs, book.indice_pair_num
            stride = pairs.shape[2]
            nmax = min(stride, max(f.shape[0], g.shape[0]))
            gw = torch.empty((K, ci, co), dtype=torch.float32, device=dev)
            nbytes = int(L.call("dbev_spconv_backward_weight_workspace_bytes", K, ci, co, nmax))
            ws = torch.empty((max(nbytes, 256),), dtype=torch.uint8, device=dev)
            L.call("dbev_spconv_backward_weight", L.ptr(f), L.ptr(g), L.ptr(pairs), L.ptr(nums), stride, nmax, int(bool(swap)), K,
                   ci, co, L.ptr(gw), L.ptr(ws), ws.numel(), L.stream_ptr(dev))
            gw = gw[:, :Cin, :Cout] if (ci, co) != (Cin, Cout) else gw
    return gin, gw


class _SparseConvFn(Function):
    """features [n_in, Cin] x weight [K, Cin, Cout] through table [n_out, K] -> [n_out, Cout] (dbev_spconv_forward / _backward_*)."""

    @staticmethod
    def forward(ctx, features, weight, table, n_out, book, swap):
        out = _conv_forward(features, weight, table, n_out)
        ctx.save_for_backward(features, weight)
        ctx.meta = (book, swap, n_out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        book, swap, n_out = ctx.meta
        gin, gw = _conv_backward(features, weight, grad_out, book, swap, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gin, gw, None, None, None, None


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """ops.py:107-126 with the reference's arguments (pair lists): converts the lists to the two tables and runs the kernels."""
    K = indice_pairs.shape[0]
    dev = features.device
    n_src = features.shape[0]
    nbr = torch.full((max(num_activate_out, 1), K), -1, dtype=torch.int32, device=dev)      # output row <- input row
    inv = torch.full((max(n_src, 1), K), -1, dtype=torch.int32, device=dev)                 # input row <- output row
    nums = indice_pair_num.cpu().tolist()
    src, dst = (1, 0) if inverse else (0, 1)
    for k, n in enumerate(nums):
        if n:
            nbr[indice_pairs[k, dst, :n].long(), k] = indice_pairs[k, src, :n]
            inv[indice_pairs[k, src, :n].long(), k] = indice_pairs[k, dst, :n]
    w = filters.reshape(K, filters.shape[-2], filters.shape[-1])
    # _SparseConvFn's backward takes `book.nbr if swap else book.inv` for the data gradient: hand it the inverse table either way
    book = _PairsView(indice_pairs.int().contiguous(), indice_pair_num.int().contiguous(), inv if inverse else None, None if inverse else inv)
    return _SparseConvFn.apply(features, w, nbr, num_activate_out, book, inverse)


# ---- structure.py ----------------------------------------------------------------------------------------------------
class SparseConvTensor(object):
    """structure.py:21-69"""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices.int() if indices.dtype != torch.int32 else indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = {}
        self.rulebooks = {}           # Rulebook objects (same keys as indice_dict + the automatic ones)
        self.grid = grid

    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key)

    def dense(self, channels_first=True):
        """[B, C, *spatial] (channels_first) or [B, *spatial, C]: zero where no voxel is active."""
        dev = self.features.device
        shape = [int(v) for v in self.spatial_shape]
        C = self.features.shape[1]
        ndim = len(shape)
        d3 = [1] * (3 - ndim) + shape
        idx4 = self.indices.int().contiguous()
        if ndim == 2:
            idx4 = torch.cat([idx4[:, :1], torch.zeros_like(idx4[:, :1]), idx4[:, 1:]], dim=1).contiguous()
        canvas = torch.empty([self.batch_size, C] + d3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_sparse_to_dense", L.ptr(self.features.float().contiguous()), L.ptr(idx4), idx4.shape[0], C,
                   self.batch_size, d3[0], d3[1], d3[2], L.ptr(canvas), L.stream_ptr(dev))
        canvas = canvas.view([self.batch_size, C] + shape)
        if channels_first:
            return canvas
        return canvas.permute(0, *range(2, ndim + 2), 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size


# ---- modules.py --------------------------------------------------------------------------------------------------------
class SparseModule(nn.Module):
    """place holder: modules derived from it take / return a SparseConvTensor inside SparseSequential"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """modules.py:49-150"""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.items())
        i = 0
        while i < len(mods):
            k, module = mods[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                # inference: conv -> eval BatchNorm1d [-> ReLU] runs as one kernel (the reference's fused() of modules.py:152-196)
                if isinstance(module, SparseConvolution) and i < len(mods) and fusable_norm(module, mods[i][1]):
                    norm = mods[i][1]
                    relu = i + 1 < len(mods) and isinstance(mods[i + 1][1], nn.ReLU)
                    input = module(input, fused=(norm, None, relu))
                    i += 2 if relu else 1
                    continue
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    input.features = module(input.features)
            else:
                input = module(input)
        return input


def fusable_norm(conv, norm):
    """conv -> norm can run in the convolution's epilogue: gradients off, BatchNorm1d with running statistics in eval mode."""
    return (not torch.is_grad_enabled() and isinstance(norm, nn.BatchNorm1d) and not norm.training and
            norm.running_mean is not None and not conv.conv1x1 and conv.weight.is_cuda)


def folded_norm(norm, bias):
    """eval BatchNorm1d (after an optional convolution bias) as y = x * scale + shift"""
    scale = torch.rsqrt(norm.running_var.float() + norm.eps)
    if norm.weight is not None:
        scale = scale * norm.weight.float()
    mean = norm.running_mean.float() if bias is None else norm.running_mean.float() - bias.float()
    shift = -mean * scale
    if norm.bias is not None:
        shift = shift + norm.bias.float()
    return scale, shift


class ToDense(SparseModule):
    def forward(self, x):
        return x.dense()


class RemoveGrid(SparseModule):
    def forward(self, x):
        x.grid = None
        return x


# ---- conv.py ---------------------------------------------------------------------------------------------------------------
def _fan_hwio(tensor):
    """conv.py:26-45 (_calculate_fan_in_and_fan_out_hwio)"""
    if tensor.ndimension() == 2:
        return tensor.size(-2), tensor.size(-1)
    rf = 1
    if tensor.dim() > 2:
        rf = tensor[..., 0, 0].numel()
    return tensor.size(-2) * rf, tensor.size(-1) * rf


class SparseConvolution(SparseModule):
    """conv.py:60-225; weight [*kernel_size, in_channels, out_channels]"""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None, fused_bn=False):
        super().__init__()
        assert groups == 1, "grouped sparse convolutions are not part of the reference's layers"
        lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        kernel_size, stride, padding, dilation = lst(kernel_size), lst(stride), lst(padding), lst(dilation)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim, self.in_channels, self.out_channels = ndim, in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = kernel_size, stride, padding, dilation
        self.conv1x1 = np.prod(kernel_size) == 1
        self.transposed, self.inverse, self.output_padding = transposed, inverse, lst(output_padding)
        self.groups, self.subm, self.indice_key, self.fused_bn = groups, subm, indice_key, fused_bn
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = _fan_hwio(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def _auto_key(self, input):
        # identifies the geometry on this index tensor (a submanifold convolution ignores its stride / padding)
        sp = () if self.subm else (tuple(self.stride), tuple(self.padding), self.transposed, tuple(self.output_padding))
        return ("auto", input.indices.data_ptr(), input.indices.shape[0], tuple(self.kernel_size), sp, tuple(self.dilation),
                self.subm)

    def forward(self, input, fused=None):
        """fused = (BatchNorm1d in eval mode, residual features or None, relu) folds that tail into the kernel (no autograd)."""
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        if self.conv1x1:
            assert fused is None
            features = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                features = features + self.bias
            out = SparseConvTensor(features, input.indices, input.spatial_shape, input.batch_size)
            out.indice_dict, out.rulebooks, out.grid = input.indice_dict, input.rulebooks, input.grid
            return out
        K = int(np.prod(self.kernel_size))
        w = self.weight.view(K, self.in_channels, self.out_channels)
        if self.inverse:
            datas = input.find_indice_pair(self.indice_key)
            assert datas is not None and self.indice_key is not None
            outids, out_spatial_shape = datas[1], datas[4]
            rb = input.rulebooks[self.indice_key]
            assert rb.K == K, "inverse conv must have same kernel size as its couple conv"
            table, n_out = rb.inv, rb.n_in
        else:
            auto = self._auto_key(input)
            key = self.indice_key if self.indice_key is not None else auto
            rb = input.rulebooks.get(key)
            if rb is None:
                rb = input.rulebooks.get(auto)          # the same geometry built under another (or no) indice_key
                if rb is None:
                    rb = build_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding,
                                        self.dilation, self.subm, self.transposed, self.output_padding)
                input.rulebooks[key] = input.rulebooks[auto] = rb
                if self.indice_key is not None:         # the reference's cache entry, same layout (conv.py:176-180)
                    input.indice_dict[self.indice_key] = IndiceData(rb, indices, spatial_shape)
            outids, out_spatial_shape = rb.outids, (spatial_shape if self.subm else rb.out_shape)
            table, n_out = rb.nbr, rb.n_out
        if fused is not None:
            norm, residual, relu = fused
            assert not torch.is_grad_enabled() and not norm.training
            scale, shift = folded_norm(norm, self.bias)
            out_features = _conv_forward(features, w, table, n_out, scale, shift, residual, relu)
        else:
            out_features = _SparseConvFn.apply(features, w, table, n_out, rb, self.inverse)
            if self.bias is not None:
                out_features = out_features + self.bias
        out = SparseConvTensor(out_features, outids, out_spatial_shape, batch_size)
        out.indice_dict, out.rulebooks, out.grid = input.indice_dict, input.rulebooks, input.grid
        return out


def _make(name, ndim, **fixed):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias, indice_key=indice_key)
        if fixed.get("inverse"):
            kw = dict(bias=bias, indice_key=indice_key)
        SparseConvolution.__init__(self, ndim, in_channels, out_channels, kernel_size, **kw, **fixed)
    cls = type(name, (SparseConvolution,), {"__init__": __init__, "__doc__": "conv.py:" + name})
    register_conv(name, cls)
    return cls


SparseConv2d = _make("SparseConv2d", 2)
SparseConv3d = _make("SparseConv3d", 3)
SubMConv2d = _make("SubMConv2d", 2, subm=True)
SubMConv3d = _make("SubMConv3d", 3, subm=True)
SparseConvTranspose2d = _make("SparseConvTranspose2d", 2, transposed=True)
SparseConvTranspose3d = _make("SparseConvTranspose3d", 3, transposed=True)
SparseInverseConv2d = _make("SparseInverseConv2d", 2, inverse=True)
SparseInverseConv3d = _make("SparseInverseConv3d", 3, inverse=True)


# ---- pool.py ---------------------------------------------------------------------------------------------------------
class _SparseMaxPoolFn(Function):
    """functional.py:77-92 (SparseMaxPoolFunction) on the neighbour / inverse tables: out = max(0, max over the paired inputs)."""

    @staticmethod
    def forward(ctx, features, table, n_out, book):
        dev = L.require_cuda(features)
        f = features.float().contiguous()
        C = f.shape[1]
        cp = (C + 3) // 4 * 4
        if cp != C:
            f = torch.nn.functional.pad(f, (0, cp - C))
        out = torch.zeros((n_out, cp), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.call("dbev_spconv_maxpool_forward", L.ptr(f), L.ptr(table), n_out, table.shape[1], cp, L.ptr(out), L.stream_ptr(dev))
        ctx.book, ctx.C = book, C
        ctx.save_for_backward(f, out)
        return out[:, :C] if cp != C else out

    @staticmethod
    def backward(ctx, grad_out):
        f, out = ctx.saved_tensors
        dev, C, cp = f.device, ctx.C, f.shape[1]
        g = grad_out.float()
        if cp != C:
            g = torch.nn.functional.pad(g, (0, cp - C))
        g = g.contiguous()
        inv = ctx.book.inv
        din = torch.zeros_like(f)
        with torch.cuda.device(dev):
            L.call("dbev_spconv_maxpool_backward", L.ptr(f), L.ptr(out), L.ptr(g), L.ptr(inv), f.shape[0], inv.shape[1], cp,
                   L.ptr(din), L.stream_ptr(dev))
        return (din[:, :C] if cp != C else din), None, None, None


def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    """functional.py:98 / ops.py:161-171 on the reference's own pair lists [K, 2, N] (-1 padded)."""
    K, n_in = indice_pairs.shape[0], features.shape[0]
    dev = features.device
    nbr = torch.full((max(num_activate_out, 1), K), -1, dtype=torch.int32, device=dev)
    inv = torch.full((max(n_in, 1), K), -1, dtype=torch.int32, device=dev)
    for k, n in enumerate(indice_pair_num.cpu().tolist()):
        if n:
            nbr[indice_pairs[k, 1, :n].long(), k] = indice_pairs[k, 0, :n]
            inv[indice_pairs[k, 0, :n].long(), k] = indice_pairs[k, 1, :n]
    book = _PairsView(indice_pairs.int().contiguous(), indice_pair_num.int().contiguous(), None, inv)
    return _SparseMaxPoolFn.apply(features, nbr, num_activate_out, book)


class SparseMaxPool(SparseModule):
    """pool.py:21-74"""

    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super().__init__()
        lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim, self.kernel_size, self.stride, self.padding = ndim, lst(kernel_size), lst(stride), lst(padding)
        self.subm, self.dilation = subm, lst(dilation)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        rb = build_rulebook(input.indices, input.batch_size, input.spatial_shape, self.kernel_size, self.stride, self.padding,
                            self.dilation, self.subm)
        out_features = _SparseMaxPoolFn.apply(input.features, rb.nbr, rb.n_out, rb)
        out = SparseConvTensor(out_features, rb.outids, input.spatial_shape if self.subm else rb.out_shape, input.batch_size)
        out.indice_dict, out.rulebooks, out.grid = input.indice_dict, input.rulebooks, input.grid
        return out


class SparseMaxPool2d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(2, kernel_size, stride, padding, dilation)


class SparseMaxPool3d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__(3, kernel_size, stride, padding, dilation)
