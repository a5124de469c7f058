"""ctypes binding of ``libretinanet_hip.so`` (C ABI: ``include/retinanet_hip.h``).

There is NO fallback: if the shared library is missing this module raises at
import, and every op refuses non-device tensors.  The product path never
touches ``oracle/``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libretinanet_hip.so")

RN_F32, RN_BF16, RN_F16 = 0, 1, 2
RN_MAX_LEVELS = 8


class RnLevel(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("stride", C.c_int32), ("num_cell", C.c_int32)]


class RnLossParams(C.Structure):
    _fields_ = [("alpha", C.c_float), ("gamma", C.c_float), ("beta", C.c_float),
                ("logit_shift", C.c_float), ("log_eps", C.c_float), ("reg_w", C.c_float * 4)]


class RnDetectParams(C.Structure):
    _fields_ = [("score_thr", C.c_float), ("min_box", C.c_float), ("nms_thr", C.c_float),
                ("max_det", C.c_int32), ("reg_w", C.c_float * 4)]


class RnPwConv(C.Structure):
    _fields_ = [("M", C.c_int64), ("Cin", C.c_int32), ("N", C.c_int32), ("taps", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("dtype", C.c_int32)]            # RN_BF16 / RN_F16 (0 = RN_BF16: older callers)


class RnPwPrologue(C.Structure):
    _fields_ = [("kind", C.c_int32), ("relu_mode", C.c_int32), ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p),
                ("fa", C.c_void_p), ("fb", C.c_void_p), ("x2", C.c_void_p), ("bits", C.c_void_p)]


class RnPwEpilogue(C.Structure):
    _fields_ = [("kind", C.c_int32), ("partial", C.c_void_p), ("resid", C.c_void_p), ("rbits", C.c_void_p), ("zprev", C.c_void_p),
                ("ea", C.c_void_p), ("eb", C.c_void_p), ("emean", C.c_void_p), ("einv", C.c_void_p),
                ("res_stride", C.c_int32), ("res_h", C.c_int32), ("res_w", C.c_int32), ("relu", C.c_int32), ("bias", C.c_void_p)]


RN_PW_PRO_NONE, RN_PW_PRO_AFFINE_RELU, RN_PW_PRO_BN_BWD = 0, 1, 2
RN_PW_EPI_NONE, RN_PW_EPI_STATS, RN_PW_EPI_RESID, RN_PW_EPI_RELU_BWD, RN_PW_EPI_BIAS = 0, 1, 2, 4, 8

_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/retinanet_hip.h one to one
SIGNATURES = {
    "rn_version": (C.c_int, []),
    "rn_hipgraph_node_census": (C.c_int, [_vp, C.POINTER(_i64)]),
    "rn_hipgraph_replace_memset_nodes": (C.c_int, [_vp, C.POINTER(_i64)]),
    "rn_status_string": (C.c_char_p, [C.c_int]),
    "rn_anchors_count": (_i64, [C.POINTER(RnLevel), C.c_int]),
    "rn_anchors_emit": (C.c_int, [C.POINTER(RnLevel), C.c_int, C.POINTER(_vp), C.c_double, _vp, _vp]),
    "rn_iou_match": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, _i64, _f32, _f32, _vp, _vp, _vp]),
    "rn_iou_match_ex": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, _i64, _f32, _f32, _vp, _vp, _i64, _vp]),
    "rn_iou_match_special_bytes": (_sz, [C.c_int, _i64]),
    "rn_iou_match_special": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, _i64, _f32, _f32, _vp, _vp, _vp, _i64, _vp]),
    "rn_iou_match_special_ex": (C.c_int, [_vp, _i64, _vp, _vp, C.c_int, _i64, _f32, _f32, _vp, _vp, _vp, _i64, C.c_int, _vp]),
    "rn_loss_workspace_bytes": (_sz, [C.c_int, _i64, C.c_int]),
    "rn_loss_fwd_bwd_levels_ex": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.c_int, C.c_int, C.c_int, C.c_int,
                                            _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(RnLossParams), _vp,
                                            C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp, _vp, _vp]),
    "rn_loss_fwd_bwd_levels_fin": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.c_int, C.c_int, C.c_int, C.c_int,
                                             _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(RnLossParams), _vp,
                                             C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp, _vp, _vp, _vp]),
    "rn_loss_fwd_bwd_levels_rp": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.c_int, C.c_int, C.c_int, C.c_int,
                                            _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(RnLossParams), _vp, C.c_int, _vp,
                                            C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp, _vp, _vp, _vp]),
    "rn_loss_match_state_bytes": (_sz, [C.c_int]),
    "rn_loss_match_fwd_bwd_levels": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.c_int, C.c_int, C.c_int, C.c_int,
                                               _vp, _i64, _vp, _vp, _vp, C.c_int, _f32, _f32, _vp, _vp, C.POINTER(RnLossParams), _vp,
                                               C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "rn_loss_fwd_bwd": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _i64, C.c_int, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                  C.POINTER(RnLossParams), _vp, _vp, _vp, _vp, _sz, _vp]),
    "rn_loss_fwd_bwd_levels": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.c_int, C.c_int, C.c_int, C.c_int,
                                         _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(RnLossParams), _vp,
                                         C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp]),
    "rn_loss_fwd_bwd_levels_timed": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_i64), C.c_int, C.c_int, C.c_int, C.c_int,
                                               _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(RnLossParams), _vp,
                                               C.POINTER(_vp), C.POINTER(_vp), _vp, _sz, _vp, _vp, _vp]),
    "rn_scale_inplace": (C.c_int, [_vp, C.c_int, _i64, _vp, _vp]),
    "rn_scale_inplace_batched": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "rn_bn_workspace_bytes": (_sz, [C.c_int]),
    "rn_bn_act_forward": (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, _f32, _f32, C.c_int,
                                    _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rn_bn_act_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                     _vp, _vp, _vp, _vp, _sz, _vp]),
    "rn_bn_stats": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rn_bn_stats_finalize": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp]),
    "rn_bn_apply": (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp, C.c_int, _vp, _vp]),
    "rn_bn_apply_res_affine": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp]),
    "rn_bn_bwd_reduce": (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rn_bn_bwd_finalize": (C.c_int, [_vp, C.c_int, _i64, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "rn_bn_bwd_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp, _vp, C.c_int, _vp]),
    "rn_pw_walkers": (C.c_int, [_i64]),
    "rn_pw_conv_forward": (C.c_int, [C.POINTER(RnPwConv), _vp, _vp, _vp, C.POINTER(RnPwPrologue), C.POINTER(RnPwEpilogue), _vp]),
    "rn_pw_wgrad_workspace_bytes": (_sz, [C.POINTER(RnPwConv)]),
    "rn_pw_conv_wgrad": (C.c_int, [C.POINTER(RnPwConv), _vp, _vp, _vp, C.POINTER(RnPwPrologue), C.POINTER(RnPwPrologue), _vp, _sz, _vp]),
    "rn_decode_clip": (C.c_int, [_vp, C.c_int, C.c_int, _i64, _vp, _i64, _vp, C.POINTER(_f32), _vp, _vp]),
    "rn_bias_act_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, _i64, C.c_int, _vp]),
    "rn_bias_act_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _i64, C.c_int, _i64, C.c_int, _vp, _sz, _vp]),
    "rn_conv3x3_canvas": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_conv3x3_canvas_sum2": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_conv3x3_canvas_batched": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _i64, _i64, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _vp]),
    "rn_maxpool3x3s2_forward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_bn_relu_maxpool3x3s2_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_maxpool3x3s2_backward": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_maxpool3x3s2_backward_bn_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rn_maxpool3x3s2_backward_bn": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_sgd_master_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32, _f32, _f32, C.c_int, C.c_int, _vp]),
    "rn_sgd_master_step_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _f32, _f32, _f32, C.c_int, C.c_int, _vp, _vp, _vp]),
    "rn_conv3x3_canvas_to_levels": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              _vp, _vp]),
    "rn_conv3x3_levels_to_canvas": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, _vp, _vp]),
    "rn_conv3x3_levels_to_canvas_relu": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, _vp, _vp, _sz, _vp]),
    "rn_conv3x3_levels_wgrad": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                          _sz, _vp]),
    "rn_conv3x3_canvas_batched_ex": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_conv3x3_dgrad_weight_batched": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_conv3x3_dgrad_weight_many": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "rn_fpn_add_upsample2x": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_fpn_upsample2x_backward": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_canvas_pack": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_colsum_rows_workspace_bytes": (_sz, [C.c_int, C.c_int]),
    "rn_colsum_rows": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _sz, _vp]),
    "rn_conv3x3_colsum_workspace_bytes": (_sz, [C.c_int, _i64, C.c_int]),
    "rn_conv3x3_canvas_dgrad_relu_batched": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _i64, _i64, C.c_int, C.c_int, C.c_int,
                                                       _vp, _sz, _vp]),
    "rn_conv3x3_wgrad_workspace_bytes": (_sz, [C.c_int, _i64]),
    "rn_conv3x3_canvas_wgrad_batched": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _sz, _vp]),
    "rn_conv3x3_dense_batched": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "rn_conv3x3_dense_band": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rn_conv3x3_dense_band_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rn_conv3x3_dense_band_stats": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rn_conv3x3_dense_splitk_workspace_bytes": (_sz, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rn_conv3x3_dense_splitk": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _sz, _vp]),
    "rn_conv3x3_dense_batched_act": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "rn_conv3x3_dense_wgrad_workspace_bytes": (_sz, [C.c_int]),
    "rn_conv3x3_dense_wgrad_batched": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _sz, _vp]),
    "rn_stem_padded_bytes": (_sz, [C.c_int, C.c_int, C.c_int]),
    "rn_stem_partial_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "rn_stem_conv_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_stem_wgrad_workspace_bytes": (_sz, [C.c_int, C.c_int, C.c_int]),
    "rn_stem_conv_wgrad": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _sz, _vp]),
    "rn_stem_conv_wgrad_bn": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _sz, _vp]),
    "rn_conv3x3_wgrad_narrow_workspace_bytes": (_sz, [C.c_int, C.c_int]),
    "rn_conv3x3_wgrad_narrow": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _sz, _vp]),
    "rn_conv3x3_narrow_forward": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "rn_copy_many": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "rn_cast_many_to_f32": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "rn_transpose_many": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "rn_conv3x3_levels_dgrad_weight": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "rn_pw_conv_wgrad_partial": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "rn_pw_block_out_conv1_walkers": (C.c_int, [_i64, C.c_int, C.c_int]),
    "rn_pw_block_out_conv1": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int] + [_vp] * 12),
    "rn_pw_dgrad_resid_sums_walkers": (C.c_int, [_i64, C.c_int, C.c_int]),
    "rn_pw_dgrad_resid_sums": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rn_pw_conv3_forward_walkers": (C.c_int, [_i64, C.c_int, C.c_int]),
    "rn_pw_conv3_forward": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rn_pw_conv3_backward_walkers": (C.c_int, [_i64, C.c_int, C.c_int]),
    "rn_pw_conv3_backward_workspace_bytes": (_sz, [_i64, C.c_int, C.c_int]),
    "rn_pw_conv3_backward": (C.c_int, [_i64, C.c_int, C.c_int, C.c_int] + [_vp] * 15 + [_sz, _vp, _vp]),
    "rn_pw_wgrad_reduce_many": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "rn_pw_wgrad_reduce_many_dt": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "rn_transform_batch": (C.c_int, [_vp, _vp, _vp, C.c_int, C.POINTER(_f32), C.POINTER(_f32), C.c_int, C.c_int, _vp,
                                     C.c_int, C.c_int, _vp]),
    "rn_nms_workspace_bytes": (_sz, [_i64, C.c_int]),
    "rn_nms_segments": (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, _f32, _vp, _vp, _vp, _sz, _vp]),
    "rn_detect_workspace_bytes": (_sz, [C.c_int, _i64, C.c_int, _i64]),
    "rn_detect": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _i64, C.c_int, _vp, _i64, _vp, C.POINTER(RnDetectParams),
                            _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "rn_detect_levels": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _i64, _vp,
                                   C.POINTER(RnDetectParams), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the MI355X HIP library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C pytorch_retinanet_amd/csrc`). "
            "There is no CPU fallback for the dense-head path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI/header mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class RetinanetHipError(RuntimeError):
    pass


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib.rn_status_string(int(status))
        raise RetinanetHipError(f"{what} failed: status {status} ({msg.decode() if msg else '?'})")
