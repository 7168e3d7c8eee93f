"""ctypes binding of libksmi.so (the C-ABI declared in include/ksmi.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent, importing this module raises.  Struct layouts mirror include/ksmi.h 1:1.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libksmi.so")

KSMI_F32, KSMI_BF16 = 0, 1
MAX_SRC, MAX_CHUNKS = 6, 72
ABI_VERSION = 7


class KsmiError(RuntimeError):
    pass


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("C", C.c_int32), ("c_off", C.c_int32), ("c_len", C.c_int32), ("relu", C.c_int32)]


class Dst(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("c_off", C.c_int32), ("n_begin", C.c_int32),
                ("n_len", C.c_int32), ("accumulate", C.c_int32), ("pad_", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("src", Src * MAX_SRC), ("dst", Dst * MAX_SRC), ("nsrc", C.c_int32), ("ndst", C.c_int32),
                ("wpk", C.c_void_p), ("bias", C.c_void_p), ("stats", C.c_void_p),
                ("mask_src", C.c_void_p), ("m_mean", C.c_void_p), ("m_rstd", C.c_void_p),
                ("m_scale", C.c_void_p), ("m_shift", C.c_void_p),
                ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("TH", C.c_int32), ("TW", C.c_int32), ("N", C.c_int32), ("Npad", C.c_int32),
                ("nchunks", C.c_int32), ("ps_cout", C.c_int32),
                ("chunk_c0", C.c_uint16 * MAX_CHUNKS), ("chunk_src", C.c_uint8 * MAX_CHUNKS),
                ("pad_x", C.c_int32), ("out_sy", C.c_int32), ("out_sx", C.c_int32), ("out_oy", C.c_int32),
                ("out_ox", C.c_int32), ("out_H", C.c_int32), ("out_W", C.c_int32),
                ("uniform_kc", C.c_int32), ("in_sy", C.c_int32), ("in_sx", C.c_int32), ("in_oy", C.c_int32), ("in_ox", C.c_int32),
                ("in_H", C.c_int32), ("in_W", C.c_int32), ("alpha", C.c_float), ("relu_out", C.c_int32), ("residC", C.c_int32),
                ("resid", C.c_void_p), ("stats_rows", C.c_int32),
                ("gate_src", C.c_void_p), ("xhat_src", C.c_void_p), ("g_mean", C.c_void_p), ("g_rstd", C.c_void_p),
                ("dir", C.c_int32), ("pad2_", C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p),
                ("nchunks", C.c_int32), ("taps", C.c_int32), ("N", C.c_int32), ("Npad", C.c_int32), ("n_mod", C.c_int32),
                ("sK", C.c_int64), ("sN", C.c_int64), ("sD", C.c_int64), ("sT", C.c_int64),
                ("flip", C.c_int32),
                ("k_off", C.c_int32 * MAX_CHUNKS), ("k_len", C.c_int32 * MAX_CHUNKS),
                ("use_tap_map", C.c_int32), ("tap_map", C.c_int32 * 16), ("uniform_kc", C.c_int32), ("k_total", C.c_int32)]


class TiffInfo(C.Structure):
    """ksmi_tiff_info (include/ksmi.h)"""
    _fields_ = [(k, C.c_int32) for k in ("width", "height", "bands", "bits", "sample_format", "compression", "predictor", "tiled",
                                         "big_endian", "bigtiff", "has_geo", "has_nodata")] + [
        ("pixel_scale", C.c_double * 2), ("origin", C.c_double * 2), ("tie_pixel", C.c_double * 2), ("nodata", C.c_double)]


class RowsumDesc(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("K", C.c_int32), ("k", C.c_int32),
                ("Cstride", C.c_int32), ("C", C.c_int32), ("accumulate", C.c_int32), ("head", C.c_int32), ("next", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [("src", Src * MAX_SRC), ("nsrc", C.c_int32),
                ("dy", C.c_void_p), ("dyC", C.c_int32), ("dy_c_off", C.c_int32),
                ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32),
                ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("TH", C.c_int32), ("TW", C.c_int32), ("N", C.c_int32), ("nchunks", C.c_int32), ("nsplit", C.c_int32),
                ("partial", C.c_void_p), ("grad", C.c_void_p),
                ("gK", C.c_int64), ("gN", C.c_int64), ("gT", C.c_int64), ("accumulate", C.c_int32),
                ("k_off", C.c_int32 * MAX_CHUNKS), ("k_len", C.c_int32 * MAX_CHUNKS),
                ("chunk_c0", C.c_uint16 * MAX_CHUNKS), ("chunk_src", C.c_uint8 * MAX_CHUNKS),
                ("uniform_kc", C.c_int32), ("k_total", C.c_int32),
                ("in_sy", C.c_int32), ("in_sx", C.c_int32), ("in_oy", C.c_int32), ("in_ox", C.c_int32), ("in_H", C.c_int32), ("in_W", C.c_int32),
                ("pad_x_set", C.c_int32), ("pad_x", C.c_int32), ("use_tap_off", C.c_int32), ("tap_off", C.c_int32 * 16),
                ("bias_grad", C.c_void_p), ("bias_accumulate", C.c_int32)]


class Op(C.Structure):
    """ksmi_op (include/ksmi.h): one entry of a compiled launch list"""
    _fields_ = [(k, C.c_int32) for k in ("kind", "sig", "lane", "side", "tag", "a", "b", "nargs")] + [("fn", C.c_void_p), ("args", C.c_void_p)]


OP_CALL, OP_ORDER, OP_WAIT_SIDE = 0, 1, 2

_vp, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_P4 = C.c_void_p * 4

# name -> (restype, argtypes).  Every symbol of include/ksmi.h is listed; load fails if one is missing.
SIGNATURES = {
    "ksmi_abi_version": (_i, []),
    "ksmi_last_kernels": (_i, [C.c_char_p, _i]),
    "ksmi_hbm_probe": (_i, [_i, _vp, _vp, _vp, C.c_size_t, _vp, _vp]),
    "ksmi_last_error": (C.c_char_p, []),
    "ksmi_set_knob": (_i, [C.c_char_p, C.c_char_p]),
    "ksmi_thunk_id": (_i, [C.c_char_p]),
    "ksmi_runner_create": (_vp, []),
    "ksmi_runner_destroy": (_i, [_vp]),
    "ksmi_runner_set_streams": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "ksmi_run_list": (_i, [_vp, _vp, _i, _i, _vp, C.POINTER(C.c_int32)]),
    "ksmi_runner_join": (_i, [_vp]),
    "ksmi_conv_dispatch_info": (_i, [C.POINTER(ConvDesc), _i, C.POINTER(C.c_int32)]),
    "ksmi_chunk_elems": (_i, [_i]),
    "ksmi_conv_grid_m": (_i, [C.POINTER(ConvDesc)]),
    "ksmi_conv_forward": (_i, [C.POINTER(ConvDesc), _i, _vp]),
    "ksmi_pack_weights": (_i, [C.POINTER(PackDesc), _i, _vp]),
    "ksmi_pack_weights_batched": (_i, [_vp, _i, _i, _vp]),
    "ksmi_conv_wgrad_workspace": (_sz, [C.POINTER(WgradDesc), _i]),
    "ksmi_conv_wgrad": (_i, [C.POINTER(WgradDesc), _i, _vp]),
    "ksmi_conv_wgrad_fuses_bias": (_i, [C.POINTER(WgradDesc), _i]),
    "ksmi_conv_stats_rows": (_i, [C.POINTER(ConvDesc), _i]),
    "ksmi_conv_gate_supported": (_i, [C.POINTER(ConvDesc), _i]),
    "ksmi_desc_size": (C.c_size_t, [_i]),
    "ksmi_conv_first_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_conv_first_forward_raw": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "ksmi_conv_first_stats_rows": (_i, [_i, _i, _i]),
    "ksmi_im2col3x3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_im2col3x3_raw": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "ksmi_conv_first_wgrad": (_i, [_vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_conv_first_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "ksmi_bn_finalize": (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "ksmi_bn_add_relu": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "ksmi_bn_fused_supported": (_i, [_i, _i, _i]),
    "ksmi_bn_fused_max_rows": (_i, []),
    "ksmi_bn_fin_add_relu": (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_bn_bwd_fin_apply_gated": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i64, _i, _i, _vp]),
    "ksmi_bnrelu_bwd_fin_apply": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i64, _i, _i, _vp]),
    "ksmi_bn_bwd_fin_apply_add": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _d, _i64, _i, _i, _vp]),
    "ksmi_bnrelu_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "ksmi_reduce_rows": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "ksmi_reduce_rows_batched": (_i, [_vp, _i, _vp]),
    "ksmi_reduce_rows_batched_wide": (_i, [_vp, _i, _i, _vp]),
    "ksmi_bnrelu_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i64, _i, _i, _vp]),
    "ksmi_bn_bwd_apply_gated": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i64, _i, _i, _vp]),
    "ksmi_bn_bwd_apply_add": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _d, _i64, _i, _i, _vp]),
    "ksmi_channel_sum": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp]),
    "ksmi_colsum": (_i, [_vp, _i64, _i, _vp, _i, _i, _vp]),
    "ksmi_gather_rows": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_scatter_rows": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_batch_sum": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp]),
    "ksmi_mse_workspace": (C.c_size_t, []),
    "ksmi_mse_loss": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i64, _i, _vp]),
    "ksmi_maxpool2x2_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_maxpool2x2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_ecam_pool": (_i, [_P4, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_ecam_pool_workspace": (_sz, [_i, _i, _i]),
    "ksmi_ecam_mlp": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ksmi_ecam_final_forward": (_i, [_P4, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_ecam_final_backward_reduce": (_i, [_P4, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_ecam_bwd_workspace": (_sz, [_i, _i, _i, _i]),
    "ksmi_ecam_mlp_backward": (_i, [_vp] * 18 + [_i, _i, _vp]),
    "ksmi_ecam_mlp_bwd_workspace": (_sz, [_i, _i]),
    "ksmi_ecam_final_backward_dx": (_i, [_P4, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_loss_workspace": (_sz, [_i, _i]),
    "ksmi_ce_dice_forward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "ksmi_ce_dice_backward": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ksmi_argmax_confusion": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_argmax_confusion_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _vp]),
    "ksmi_adamw_step": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _vp]),
    "ksmi_adam_step_mirror": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "ksmi_sgd_step": (_i, [_vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _vp]),
    "ksmi_layernorm_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "ksmi_layernorm_bwd_blocks": (_i, [_i]),
    "ksmi_layernorm_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "ksmi_gelu_forward": (_i, [_vp, _vp, _i64, _i, _vp]),
    "ksmi_gelu_backward": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "ksmi_add": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "ksmi_relu_backward": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "ksmi_relu_forward": (_i, [_vp, _vp, _i64, _i, _vp]),
    "ksmi_sar_preprocess": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, C.c_float, _vp]),
    "ksmi_bmm_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_float, _i, _vp]),
    "ksmi_softmax_rows_f32": (_i, [_vp, _vp, _i64, _i, C.c_float, _vp]),
    "ksmi_softmax_rows_backward_f32": (_i, [_vp, _vp, _vp, _i64, _i, C.c_float, _vp]),
    "ksmi_semantic_tokens_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_semantic_tokens_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_token_cross_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "ksmi_token_cross_bwd_workspace": (_sz, [_i, _i, _i]),
    "ksmi_token_cross_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "ksmi_tiff_info_read": (_i, [C.c_char_p, C.POINTER(TiffInfo)]),
    "ksmi_tiff_read_f32": (_i, [C.c_char_p, _vp, _i64, C.POINTER(TiffInfo)]),
    "ksmi_tiff_read_native": (_i, [C.c_char_p, _vp, _i64, C.POINTER(TiffInfo)]),
    "ksmi_tile_batch_read": (_i, [C.POINTER(C.c_char_p), _i, _vp, _i, _i, _i]),
    "ksmi_tile_batch_read_bands": (_i, [C.POINTER(C.c_char_p), _i, _vp, _i, _i, _i, _i]),
    "ksmi_tiles_fill_nodata": (_i, [_vp, _i, _i, _i, _i]),
    "ksmi_cast_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "ksmi_up_gemm_supported": (_i, [_i, _i, _i, _i, _i]),
    "ksmi_up_wgrad_supported": (_i, [_i, _i, _i, _i, _i]),
    "ksmi_up_pack_weight": (_i, [_vp, _vp, _i, _vp]),
    "ksmi_up_pack_weights_batched": (_i, [_vp, _vp, _vp, _i, _vp]),
    "ksmi_up_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_up_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_up_wgrad_workspace": (C.c_size_t, [_i, _i, _i, _i]),
    "ksmi_up_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_gemm_nt": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_gemm_nn": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_im2col": (_i, [_vp, _vp] + [_i] * 11 + [_i, _i, _vp]),
    "ksmi_col2im": (_i, [_vp, _vp, _i] + [_i] * 11 + [_i, _vp]),
    "ksmi_im2col_tc": (_i, [_vp, _vp] + [_i] * 11 + [_i, _vp]),
    "ksmi_col2im_tc": (_i, [_vp, _vp, _i] + [_i] * 11 + [_i, _vp]),
    "ksmi_weight_to_tc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_grad_from_tc": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_maxpool3x3s2_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_maxpool3x3s2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_maxpool3x3s2_forward_idx": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_maxpool3x3s2_backward_idx": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_affine": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, C.c_float, _i, _vp]),
    "ksmi_dwconv3x3_gelu_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_dwconv3x3_gelu_forward_drop": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_uint32, C.c_float, C.c_uint32, _vp, _i, _vp]),
    "ksmi_gelu_backward_drop": (_i, [_vp, _vp, _vp, _i64, C.c_uint32, C.c_float, C.c_uint32, _vp, _i, _vp]),
    "ksmi_dwconv3x3_backward_input": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_dwconv3x3_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_sr_attention_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "ksmi_sr_attention_bwd_workspace": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "ksmi_sr_attention_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "ksmi_attention_bwd_workspace": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "ksmi_rng_advance": (_i, [_vp, _vp]),
    "ksmi_bn_relu_drop2d": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, C.c_uint32, C.c_float, C.c_uint32, _vp, _i, _vp]),
    "ksmi_reduce_rows_scaled": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, C.c_float, _vp]),
    "ksmi_bnrelu_bwd_apply_scaled": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i64, _i, C.c_float, _i, _vp]),
    "ksmi_absdiff_forward": (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    "ksmi_absdiff_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp]),
    "ksmi_dropout_apply": (_i, [_vp, _vp, _vp, _i64, _i, _i, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, _vp, _i, _vp]),
    "ksmi_sr_attention_forward_drop": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_uint32, C.c_float, C.c_uint32, _vp, _i, _vp]),
    "ksmi_sr_attention_backward_drop": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.c_float, C.c_uint32, C.c_float, C.c_uint32,
                                             _vp, _i, _vp]),
    "ksmi_bilinear_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_bilinear_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_bn_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp]),
    "ksmi_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_double, _i64, _i, _i, _vp]),
    "ksmi_out_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i64, _i, _i, _vp]),
    "ksmi_dout_to_nhwc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i, _i, _vp]),
    "ksmi_drop_cls": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ksmi_logits_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i64, _i, _vp]),
    "ksmi_dlogits_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i64, _i, _vp]),
    "ksmi_patchify": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_vit_embed_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_vit_embed_backward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_attention_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ksmi_attention_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp]),
    "ksmi_upsample2_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_upsample2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ksmi_fill_zero": (_i, [_vp, _sz, _vp]),
    "ksmi_nchw_to_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_nhwc_to_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ksmi_selftest_mma": (_i, [_vp, _vp, _vp, _i, _vp]),
    "ksmi_selftest_tr16": (_i, [_vp, _vp, _vp]),
}

_lib = None


def load():
    """dlopen libksmi.so and bind every declared symbol.  Raises KsmiError on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KsmiError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C kurosiwo_amd/csrc`.")
    try:
        import torch  # noqa: F401  (libamdhip64 is resolved from the torch process first)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KsmiError(f"libksmi.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ksmi_abi_version() != ABI_VERSION:
        raise KsmiError(f"libksmi ABI {lib.ksmi_abi_version()} != binding {ABI_VERSION}")
    for which, cls in enumerate((ConvDesc, WgradDesc, PackDesc, RowsumDesc, TiffInfo)):        # the ctypes mirrors match the compiled structs
        if lib.ksmi_desc_size(which) != C.sizeof(cls):
            raise KsmiError(f"libksmi {cls.__name__}: {lib.ksmi_desc_size(which)} bytes in the library, {C.sizeof(cls)} in the binding")
    _lib = lib
    return lib


def set_knob(name, value):
    """ksmi_set_knob (include/ksmi.h): a run-time switch of the launchers (tests / probes only); value None = built-in default"""
    rc = load().ksmi_set_knob(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        check(rc, "ksmi_set_knob")


class knobs:
    """with knobs(KSMI_IGEMM4_CUS="3", KSMI_IGEMM4_VAR="8,2"): ...  -- restores the built-in defaults on exit"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            set_knob(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            set_knob(k, None)
        return False


def check(rc, what=""):
    if rc != 0:
        msg = load().ksmi_last_error()
        raise KsmiError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
