"""ctypes binding of libdbev_hip.so (C ABI declared in include/dbev_hip.h).

The product path has NO CPU fallback: if the library is missing or was not built,
``lib()`` raises.  Device pointers are passed as integers (``tensor.data_ptr()``);
the HIP stream is torch's *current* stream so the kernels order correctly with the
surrounding PyTorch-ROCm ops (the reference launched on the default stream,
bev_pool_cuda.cu:88 -- noted in SURVEY 8b).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBEV_HIP_LIB") or os.path.join(_HERE, "libdbev_hip.so")   # override: A/B builds
_lib = None

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_d = ctypes.c_double
_sz = ctypes.c_size_t
_ll = ctypes.c_longlong

# name -> argtypes (restype is int unless listed in _RESTYPES).  Must list EVERY symbol
# include/dbev_hip.h declares (tests/test_abi.py cross-checks against the header).
_SIGNATURES = {
    "dbev_abi_version": [],
    "dbev_target_arch": [],
    "dbev_kernel_timing_enable": [_i],
    "dbev_kernel_timing_read": [_p, _p, _p, _i],
    "dbev_kernel_name": [_i],
    "dbev_fallback_note": [_i],
    "dbev_fallback_count": [_i],
    "dbev_fallback_reset": [],
    "dbev_bev_pool_forward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "dbev_bev_pool_backward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "dbev_bev_pool_prepare": [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_bev_pool_prepare_i64": [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_transpose_bcs_to_bsc": [_p, _p, _i, _i, _i, _p],
    "dbev_dynamic_voxelize": [_p, _p, _i, _i, _p, _p, _i, _p],
    "dbev_hard_voxelize_workspace_bytes": [_i, _p, _p],
    "dbev_hard_voxelize": [_p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _p, _sz, _p],
    "dbev_dynamic_scatter_workspace_bytes": [_i, _i, _i, _i],
    "dbev_dynamic_scatter_prepare": [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_dynamic_scatter_reduce": [_p, _p, _p, _p, _i, _i, _i, _p],
    "dbev_dynamic_scatter_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "dbev_pillars_scatter": [_p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p],
    "dbev_pillars_scatter_backward": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "dbev_lift_splat_workspace_bytes": [_i, _i],
    "dbev_lift_splat_prepare": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_lift_splat_prepare_cam": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_lift_splat_forward": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dbev_lift_splat_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dbev_splat_forward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "dbev_splat_backward": [_p, _p, _p, _i, _i, _p],
    "dbev_fg_scale_mask": [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p],
    "dbev_abs_mean_maps_workspace_bytes": [_i, _i, _i],
    "dbev_abs_mean_maps": [_p, _i, _i, _i, _p, _p, _p, _sz, _p],
    "dbev_fgd_masked_mse_workspace_bytes": [_i, _i, _i],
    "dbev_fgd_masked_mse_forward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _sz, _p],
    "dbev_fgd_masked_mse_backward": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p],
    "dbev_pillar_vfe_workspace_bytes": [_i, _i, _i, _i],
    "dbev_pillars_canvas": [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dbev_pillar_vfe_canvas": [_p, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p, _i, _p, _sz, _p],
    "dbev_upsample_bilinear_ac_forward": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "dbev_upsample_bilinear_ac_backward": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "dbev_dcnv2_im2col": [_p, _p, _p] + [_i] * 11 + [_p],
    "dbev_dcnv2_col2im_workspace_bytes": [_i] * 8,
    "dbev_dcnv2_col2im": [_p, _p, _p, _p, _p] + [_i] * 11 + [_p, _sz, _p],
    "dbev_abs_mean_maps_nhwc_workspace_bytes": [_i, _i, _i],
    "dbev_abs_mean_maps_nhwc": [_p, _i, _i, _i, _p, _p, _p, _p, _sz, _p],
    "dbev_fgd_masked_mse_nhwc_workspace_bytes": [_i, _i, _i],
    "dbev_fgd_masked_mse_forward_nhwc": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _sz, _p],
    "dbev_fgd_masked_mse_backward_nhwc": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p],
    "dbev_centerhead_targets": [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _f, _f, _f, _f, _f, _i, _i, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_centerhead_loss_workspace_bytes": [_i, _i, _i, _i, _i],
    "dbev_centerhead_loss_forward": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _sz, _p],
    "dbev_centerhead_loss_backward": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p],
    "dbev_skinny_conv3x3_workspace_bytes": [_i, _i],
    "dbev_skinny_conv3x3_forward": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dbev_skinny_conv3x3_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p],
    "dbev_skinny_conv3x3_multi_workspace_bytes": [_i, _i],
    "dbev_skinny_conv3x3_multi_forward": [_p, _ll, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dbev_skinny_conv3x3_multi_backward": [_p, _p, _ll, _p, _p, _ll, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p],
    "dbev_grid_sample_bilinear_nhwc": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "dbev_maxpool3x3s2_forward": [_p, _i, _i, _i, _i, _p, _p, _p],
    "dbev_maxpool3x3s2_backward": [_p, _p, _i, _i, _i, _i, _p, _p],
    "dbev_norm_relu_maxpool3x3s2_forward": [_p, _p, _i, _i, _i, _i, _p, _p, _p],
    "dbev_stem_pool_norm_backward_workspace_bytes": [_i, _i, _i, _i],
    "dbev_stem_pool_norm_backward": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p],
    "dbev_stem7x7s2_stats_rows": [_i, _i, _i],
    "dbev_stem7x7s2_workspace_bytes": [_i, _i, _i],
    "dbev_stem7x7s2_forward": [_p, _p, _i, _i, _i, _p, _p, _p, _sz, _p],
    "dbev_stem7x7s2_backward_weight": [_p, _p, _i, _i, _i, _p, _p, _sz, _p],
    "dbev_channel_sum_workspace_bytes": [_ll, _i],
    "dbev_gemm_bf16x6_packed_bytes": [_i, _i],
    "dbev_gemm_bf16x6_pack": [_p, _ll, _ll, _i, _i, _i, _p, _p],
    "dbev_gemm_bf16x6_pack_pair": [_p, _ll, _ll, _i, _i, _i, _p, _i, _p, _p],
    "dbev_gemm_bf16x6_forward": [_p, _p, _p, _ll, _i, _i, _i, _i, _p],
    "dbev_gemm_bf16x6_forward_stats": [_p, _p, _p, _p, _ll, _i, _i, _i, _i, _p],
    "dbev_gemm_bf16x6_forward_bias": [_p, _p, _p, _p, _ll, _i, _i, _i, _i, _p],
    "dbev_conv3x3s2_bf16x6_ok": [_i, _i, _i, _i, _i],
    "dbev_conv3x3s2_bf16x6_forward_stats": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dbev_gemm_bf16x6_stats_rows": [_ll],
    "dbev_gemm_bf16x6_backward_weight_workspace_bytes": [_ll, _i, _i, _i],
    "dbev_gemm_bf16x6_backward_weight": [_p, _p, _p, _ll, _i, _i, _i, _p, _sz, _p],
    "dbev_channel_sum_nhwc": [_p, _ll, _i, _p, _p, _sz, _p],
    "dbev_wino_filter_pack_multi": [_p, _i, _ll, _p],
    "dbev_gemm_bf16x6_pack_multi": [_p, _i, _ll, _p],
    "dbev_subsample2_nhwc": [_p, _p, _i, _i, _i, _i, _p],
    "dbev_upsample2_zero_nhwc": [_p, _p, _i, _i, _i, _i, _p],
    "dbev_depth_head_forward": [_p, _p, _p, _p, _ll, _i, _i, _p, _p, _p, _p],
    "dbev_spconv_build_workspace_bytes": [_i, _i, _p, _p, _i, _i],
    "dbev_spconv_outputs": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _sz, _p],
    "dbev_spconv_neighbors": [_p, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _sz, _p],
    "dbev_spconv_forward": [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    "dbev_spconv_pair_lists_workspace_bytes": [_i, _i],
    "dbev_spconv_pair_lists": [_p, _i, _i, _i, _p, _p, _p, _sz, _p],
    "dbev_spconv_inverse_table": [_p, _i, _i, _i, _p, _p],
    "dbev_spconv_maxpool_forward": [_p, _p, _i, _i, _i, _p, _p],
    "dbev_spconv_maxpool_backward": [_p, _p, _p, _p, _i, _i, _i, _p, _p],
    "dbev_spconv_forward_fused": [_p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _p],
    "dbev_sparse_to_dense": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "dbev_spconv_backward_data": [_p, _p, _p, _i, _i, _i, _i, _p, _p, _sz, _p],
    "dbev_spconv_backward_weight_workspace_bytes": [_i, _i, _i, _i],
    "dbev_spconv_backward_weight": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _sz, _p],
    "dbev_conv1x1_stats_rows": [_ll, _i, _i],
    "dbev_conv1x1_forward": [_p, _p, _p, _p, _ll, _i, _i, _i, _p],
    "dbev_gemm1x1_stats_rows": [_ll, _i, _i, _i],
    "dbev_gemm1x1_forward": [_p, _p, _p, _p, _ll, _i, _i, _i, _p],
    "dbev_gemm1x1_backward_weight_workspace_bytes": [_ll, _i, _i, _i],
    "dbev_gemm1x1_backward_weight": [_p, _p, _p, _ll, _i, _i, _i, _p, _sz, _p],
    "dbev_wino_filter_floats": [_i, _i],
    "dbev_wino_filter_pack": [_p, _ll, _ll, _ll, _ll, _i, _i, _i, _p, _p],
    "dbev_wino_filter_pack_pair": [_p, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _p, _p, _p],
    "dbev_wino_conv3x3_stats_rows": [_i, _i, _i, _i, _i],
    "dbev_wino_conv3x3_forward_kernel": [_i, _i, _i, _i, _i],
    "dbev_wino_conv3x3_forward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "dbev_wino_conv3x3_forward_act": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "dbev_wino_conv3x3_backward_weight_workspace_bytes": [_i, _i, _i, _i, _i],
    "dbev_wino_conv3x3_backward_weight": [_p, _p, _p, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _i, _p, _sz, _p],
    "dbev_range_voxel_coords": [_p, _i, _i, _p, _p, _i, _p, _p],
    "dbev_virtual_voxel_reduce": [_p, _p, _p, _p, _i, _p],
    "dbev_msda_backward_workspace_bytes": [_i, _i, _i, _i, _i, _i],
    "dbev_msda_forward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p],
    "dbev_msda_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p],
    "dbev_adapt_mse_map_slices": [_i, _i],
    "dbev_adapt_mse_forward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p],
    "dbev_adapt_mse_backward_ds": [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p],
    "dbev_bn_act_workspace_bytes": [_ll, _i],
    "dbev_bn_act_train_forward": [_p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
    "dbev_bn_act_train_forward_pre": [_p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _ll, _i, _p, _i, _p, _sz, _p],
    "dbev_bn_act_infer": [_p, _p, _p, _p, _p, _p, _f, _i, _p, _ll, _i, _p, _sz, _p],
    "dbev_bn_act_backward": [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
    "dbev_bn_infer_coef": [_p, _p, _p, _p, _f, _i, _p, _p],
    "dbev_bn_act_apply": [_p, _p, _p, _i, _p, _ll, _i, _p],
    "dbev_bn_act_backward2": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
    "dbev_points_to_depth_maps": [_p, _i, _i, _p, _i, _i, _i, _i, _f, _f, _p, _p],
    "dbev_bn_dual_workspace_bytes": [_ll, _i],
    "dbev_bn_dual_train_forward": [_p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p,
                                   _ll, _i, _p, _sz, _p],
    "dbev_bn_dual_train_forward_pre": [_p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p,
                                       _ll, _i, _p, _i, _p, _i, _p, _sz, _p],
    "dbev_bn_dual_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
    "dbev_bn_dual_backward2": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
    "dbev_bn_act_train_forward_mask": [_p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _ll, _i, _p, _i, _p, _p, _sz, _p],
    "dbev_bn_act_backward3": [_p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
    "dbev_bn_dual_train_forward_mask": [_p, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _p,
                                        _ll, _i, _p, _i, _p, _i, _p, _p, _sz, _p],
    "dbev_bn_dual_backward3": [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _ll, _i, _p, _sz, _p],
}
_RESTYPES = {"dbev_target_arch": ctypes.c_char_p,
             "dbev_wino_filter_floats": ctypes.c_longlong,
             "dbev_stem7x7s2_workspace_bytes": ctypes.c_longlong,
             "dbev_stem_pool_norm_backward_workspace_bytes": ctypes.c_size_t,
             "dbev_gemm1x1_backward_weight_workspace_bytes": ctypes.c_size_t,
             "dbev_wino_conv3x3_backward_weight_workspace_bytes": ctypes.c_size_t,
             "dbev_fallback_count": ctypes.c_longlong,
             "dbev_kernel_name": ctypes.c_char_p,
             "dbev_msda_backward_workspace_bytes": ctypes.c_size_t,
             "dbev_spconv_build_workspace_bytes": ctypes.c_size_t,
             "dbev_spconv_pair_lists_workspace_bytes": ctypes.c_size_t,
             "dbev_spconv_backward_weight_workspace_bytes": ctypes.c_size_t,
             "dbev_kernel_timing_read": ctypes.c_int,
             "dbev_pillar_vfe_workspace_bytes": ctypes.c_size_t,
             "dbev_channel_sum_workspace_bytes": ctypes.c_size_t,
             "dbev_gemm_bf16x6_packed_bytes": ctypes.c_longlong,
             "dbev_gemm_bf16x6_backward_weight_workspace_bytes": ctypes.c_size_t,
             "dbev_abs_mean_maps_workspace_bytes": ctypes.c_size_t,
             "dbev_fgd_masked_mse_workspace_bytes": ctypes.c_size_t,
             "dbev_lift_splat_workspace_bytes": ctypes.c_size_t,
             "dbev_hard_voxelize_workspace_bytes": ctypes.c_size_t,
             "dbev_dynamic_scatter_workspace_bytes": ctypes.c_size_t,
             "dbev_bn_act_workspace_bytes": ctypes.c_size_t,
             "dbev_bn_dual_workspace_bytes": ctypes.c_size_t,
             "dbev_skinny_conv3x3_workspace_bytes": ctypes.c_size_t,
             "dbev_skinny_conv3x3_multi_workspace_bytes": ctypes.c_size_t,
             "dbev_centerhead_loss_workspace_bytes": ctypes.c_size_t,
             "dbev_dcnv2_col2im_workspace_bytes": ctypes.c_size_t,
             "dbev_abs_mean_maps_nhwc_workspace_bytes": ctypes.c_size_t,
             "dbev_fgd_masked_mse_nhwc_workspace_bytes": ctypes.c_size_t}
_NO_CHECK = set(_RESTYPES) | {"dbev_adapt_mse_map_slices", "dbev_conv1x1_stats_rows", "dbev_wino_conv3x3_stats_rows", "dbev_wino_conv3x3_forward_kernel", "dbev_gemm1x1_stats_rows", "dbev_gemm_bf16x6_stats_rows"}


class DbevHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DbevHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C distill_bev_amd/csrc` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the hot path.")
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing -> loud
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = h
    return _lib


# ---- optional live timing of individual ABI calls (bench.py roofline) ----
_timers = {}


def enable_timing(name):
    """Record a HIP-event pair (on torch's current stream = the launch stream) around
    every subsequent call of ABI entry `name`; read back with timing_ms()."""
    _timers[name] = []


def disable_timing(name=None):
    if name is None:
        _timers.clear()
    else:
        _timers.pop(name, None)


def timing_ms(name):
    """Per-call durations in ms of the recorded calls (synchronises)."""
    torch.cuda.synchronize()
    return [r[0].elapsed_time(r[1]) for r in _timers.get(name, [])]


def timing_bytes(name):
    """Algorithmic bytes the callers attached to the recorded calls (0 where none was given)."""
    return [r[2] for r in _timers.get(name, [])]


def timing_enabled(name):
    return name in _timers


def call(name, *args, alg_bytes=0):
    """Invoke ABI entry `name`, raising on a non-zero return code.  `alg_bytes`: the call's algorithmic
    HBM bytes, kept next to its event pair when timing is enabled (bench roofline bookkeeping)."""
    fn = getattr(lib(), name)
    rec = _timers.get(name)
    if rec is not None:
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        rec.append((s, e, alg_bytes))
    else:
        rc = fn(*args)
    if name in _NO_CHECK:
        return rc
    check(rc, name)
    return 0


KERNEL_IDS = {"bn_stats": 1, "bn_finalize": 2, "bn_apply": 3, "bn_apply_res": 4, "bn_bwd_reduce": 5, "bn_bwd_reduce_y": 6,
              "bn_bwd_finalize": 7, "bn_bwd_dx": 8, "bn_bwd_dx_res": 9, "sp_conv_fwd": 10, "msda_fwd": 11, "msda_bwd_sample": 12,
              "msda_gv_gather": 13, "adapt_mse_fwd": 14, "c1x1_fwd": 15, "wino_fwd": 16, "wino_wgrad": 17, "g1_fwd": 18, "g1_wgrad": 19, "b6_fwd": 20, "b6_wgrad": 21,
              "stem_fwd": 22, "stem_wgrad": 23}        # DBEV_K_* of include/dbev_hip.h


def kernel_timing(which):
    """Per-KERNEL event log of the library (dbev_kernel_timing_enable): which = False/None (off), True (every
    instrumented kernel) or an iterable of KERNEL_IDS names."""
    if which is True:
        mask = -1
    elif not which:
        mask = 0
    else:
        mask = 0
        for k in which:
            mask |= 1 << KERNEL_IDS[k]
    global _KT_MASK
    _KT_MASK = mask
    lib().dbev_kernel_timing_enable(mask)


_KT_MASK = 0


def kernel_timing_active(full=False):
    """is any host-side timing on (the library's kernel event log or the per-entry-point event brackets)?  HIP-graph capture
    (graphed.py) steps aside while it is: events are host code.  full=True: is EVERY kernel being logged (mask -1 / brackets on)?"""
    if full:
        return _KT_MASK == -1 or bool(_timers)
    return _KT_MASK != 0 or bool(_timers)


def kernel_timing_read():
    """-> {kernel name: [(ms, algorithmic_bytes), ...]} of every kernel launched by the instrumented entry points since
    the last read (synchronises on the recorded events)."""
    h = lib()
    cap = 1 << 16
    kid = (ctypes.c_int * cap)()
    ms = (ctypes.c_float * cap)()
    by = (ctypes.c_longlong * cap)()
    n = min(h.dbev_kernel_timing_read(kid, ms, by, cap), cap)
    out = {}
    for i in range(n):
        out.setdefault(h.dbev_kernel_name(kid[i]).decode(), []).append((ms[i], by[i]))
    return out


FALLBACK_SITES = {"bn_act": 0, "skinny_conv": 1, "adapt_mse": 2, "pillar_vfe": 3, "head_batch": 4, "depth_head": 5}   # DBEV_FB_*
_warned_fallbacks = set()


def note_fallback(site, why=""):
    """A device tensor took the stock torch path instead of the fused kernel of `site`: count it in the library's ledger
    (dbev_fallback_note) and warn ONCE per (site, reason)."""
    lib().dbev_fallback_note(FALLBACK_SITES[site])
    if (site, why) not in _warned_fallbacks:
        _warned_fallbacks.add((site, why))
        import warnings
        warnings.warn(f"distill_bev_amd: {site} took the stock torch path on a device tensor ({why or 'ineligible'}); "
                      "further occurrences are only counted (fallback_counts())", RuntimeWarning, stacklevel=3)


def fallback_counts():
    """-> {site: notes since load / the last reset}, plus 'total'"""
    h = lib()
    out = {k: int(h.dbev_fallback_count(v)) for k, v in FALLBACK_SITES.items()}
    out["total"] = int(h.dbev_fallback_count(-1))
    return out


def fallback_reset():
    lib().dbev_fallback_reset()


def touched(*tensors):
    """the library wrote these tensors through raw pointers (running statistics updated inside a kernel): bump their version counters so
    that everything keyed on `_version` (autograd's saved-tensor check, the kept eval coefficients of bn_act) sees the change"""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


_VERSION_HOOK = []


def ensure_param_version_hook():
    """Everything this package keeps per weight (packed Winograd filters, bf16 planes of the 1x1 filters, eval-mode norm coefficients)
    is keyed on the tensor's version counter -- and torch's FUSED optimizers (torch.optim.AdamW(fused=True), the one train_step.Trainer
    uses) update the parameters without moving it (measured: `_version` 0 -> 0 across `step()`, CPU and GPU; foreach / single-tensor
    implementations do move it).  A stale pack would make every step after the first run its forward pass on the step-0 weights.  One
    global optimizer post-step hook bumps the counters of the parameters the step had gradients for.  Idempotent."""
    if _VERSION_HOOK:
        return
    from torch.optim.optimizer import register_optimizer_step_post_hook

    def _bump(optimizer, args, kwargs):
        for group in optimizer.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    torch.autograd.graph.increment_version(p)

    _VERSION_HOOK.append(register_optimizer_step_post_hook(_bump))


# ---- debug guard of everything kept per weight ---------------------------------------------------------------------------------------
# ---- what a captured hipGraph baked in (graphed.py) ---------------------------------------------------------------------------------
# While DERIVED_LOG is a list, every accessor of a kept weight-derived buffer (wino.packed_pair, wino.folded_pack, gemm_bf6.packed,
# bn_act._eval_coef) appends (kind, owner, args, tensors) for what it hands out: the kernels captured meanwhile read those buffers
# through their raw addresses.  graphed.GraphedNoGrad keeps the tensors alive with the graph and, before every replay, calls the same
# accessor again (`revalidate`): a stale entry is re-derived IN PLACE by the accessor (same buffer -> the replay reads fresh values), a
# buffer that moved means the graph is captured again.
DERIVED_LOG = None


def note_derived(kind, owner, args, tensors):
    if DERIVED_LOG is not None:
        DERIVED_LOG.append((kind, owner, args, tuple(tensors)))


def revalidate(kind, owner, args):
    """-> the tensors the accessor of `kind` hands out NOW for (owner, args) (made fresh on the way, in place where the buffer exists)"""
    if kind == "wino_pair":
        from . import wino
        return tuple(wino.packed_pair(owner, *args))
    if kind == "wino_folded":
        from . import wino
        hit = wino.folded_pack(owner, *args)
        return (hit[2], hit[3])
    if kind == "bf6":
        from . import gemm_bf6
        return (gemm_bf6.packed(owner, *args),)
    if kind == "eval_coef":
        from . import bn_act
        return (bn_act._eval_coef(owner, *args),)
    raise DbevHipError(f"revalidate: unknown kind {kind}")


# DBEV_CHECK_PACKS=N (N >= 1): every N-th REUSE of a kept derivative of a weight -- packed Winograd filters (wino.packed_pair, the folded
# conv + norm packs), bf16 planes of a 1x1 filter (gemm_bf6.packed), eval-mode norm coefficients (bn_act._eval_coef) -- re-reads the live
# source tensors and compares their fingerprint with the one taken when the derivative was made; a mismatch means the weight was written
# without its version counter moving (a `.data` write, an EMA hook, a non-torch optimizer, a collective on `t.data`) and raises instead of
# silently running the forward on stale weights.  Costs a reduction over the weight and a host read-back per check: a debug mode.
CHECK_PACKS = int(os.environ.get("DBEV_CHECK_PACKS", "0") or 0)
_check_calls = [0]


def fingerprint(*tensors):
    """device f64[2 n]: (sum, sum of squares) of each source tensor, or None when the guard is off"""
    if not CHECK_PACKS:
        return None
    parts = []
    for t in tensors:
        if t is not None:
            d = t.detach().double()
            parts += [d.sum(), (d * d).sum()]
    return torch.stack(parts) if parts else None


def check_fingerprint(fp, what, *tensors):
    """called where a kept derivative is about to be reused; every CHECK_PACKS-th call compares `fp` with the live tensors"""
    if not CHECK_PACKS or fp is None:
        return
    _check_calls[0] += 1
    if _check_calls[0] % CHECK_PACKS:
        return
    now = fingerprint(*tensors)
    if now is None or now.shape != fp.shape or not torch.equal(fp.nan_to_num(), now.nan_to_num()):
        raise DbevHipError(f"DBEV_CHECK_PACKS: the kept {what} is STALE -- its source tensor changed without its version counter moving "
                           "(a .data write, an EMA hook, a custom optimizer?).  Write through the tensor itself (p.copy_ / "
                           "torch.autograd.graph.increment_version(p)) or call distill_bev_amd.bn_act.invalidate_eval_coef(model) after "
                           "such updates.")


def h2d(t, device):
    """CPU tensor -> device without stalling the host on the GPU queue: a copy from PAGEABLE memory blocks until everything queued
    before it has run (the host then sits idle for most of a step); staged through the pinned-memory cache the copy is asynchronous
    and the cache keeps the staging block alive until the copy has completed."""
    if torch.device(device).type != "cuda" or t.is_cuda:
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def h2d_like(ref, data, dtype=None):
    """ref.new_tensor(data) without the blocking pageable copy (see h2d)"""
    return h2d(torch.as_tensor(data, dtype=dtype or ref.dtype), ref.device)


def host_ptrs(tensors):
    """HOST array of device pointers (argument tables of the multi-tensor entry points)."""
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def host_ints(values):
    vals = [int(v) for v in values]
    return (ctypes.c_int32 * len(vals))(*vals)


def host_floats(values):
    """ctypes float array for the *_host arguments of the ABI."""
    vals = [float(v) for v in values]
    return (ctypes.c_float * len(vals))(*vals)


def exported_symbols():
    return sorted(_SIGNATURES)


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def check(rc, what):
    if rc != 0:
        raise DbevHipError(f"{what} failed with code {rc}"
                           + (" (invalid argument)" if rc == 10001 else " (hipError_t)"))


def require_cuda(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise DbevHipError("libdbev_hip ops need device (HBM) tensors; got a CPU tensor. "
                               "There is no CPU fallback in the product path.")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise DbevHipError("all tensors of one op must live on the same GPU")
    return dev


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
