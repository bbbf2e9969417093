"""ctypes binding of libbrainfm_hip.so (the C ABI in include/brainfm_hip.h).

There is no fallback: if the shared library is missing or a call is rejected,
this module raises.  Build with ``python -m brainfm_amd.build``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbrainfm_hip.so")

ERR = {0: "BFM_OK", -1: "BFM_E_ARG", -2: "BFM_E_SHAPE", -3: "BFM_E_WORKSPACE", -4: "BFM_E_LAUNCH"}

ROLE_PLAIN, ROLE_CT, ROLE_BIAS_LOG, ROLE_SEG, ROLE_DIST, ROLE_SR, ROLE_PATHOL = range(7)
(EW_EXP, EW_AFFINE, EW_CLAMP, EW_CLAMP_MIN, EW_GAMMA, EW_SIGMOID, EW_DIV, EW_NONZERO, EW_SUB_DIV, EW_GE,
 EW_NAN_TO_NUM) = range(11)
(EW_ADD, EW_MUL, EW_MUL_EXP, EW_AXPY_CLAMP0, EW_AXPY, EW_DIV2, EW_ZERO_WHERE_ZERO) = range(7)


class BfmError(RuntimeError):
    pass


class Upsample(C.Structure):
    _fields_ = [("d", C.c_int), ("h", C.c_int), ("w", C.c_int),
                ("mapD", C.c_void_p), ("mapH", C.c_void_p), ("mapW", C.c_void_p),
                ("repD", C.c_void_p), ("repH", C.c_void_p), ("repW", C.c_void_p)]


CLASS_STATS_BLOCKS = 1024     # BFM_CLASS_STATS_BLOCKS of include/brainfm_hip.h


class TailDesc(C.Structure):
    _fields_ = [("n_out", C.c_int), ("c_feat", C.c_int),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p), ("roles", C.c_void_p), ("out_slot", C.c_void_p),
                ("seg_first", C.c_int), ("n_seg", C.c_int), ("seg_lut", C.c_void_p),
                ("n_dist", C.c_int), ("dist_first", C.c_int), ("max_dist", C.c_float),
                ("unit_feat", C.c_int), ("slot_high_res", C.c_int), ("slot_fake_cortical", C.c_int),
                ("n_maps", C.c_int), ("head_wmax", C.c_float), ("skip_zero_input", C.c_int)]


class ZoomAxis(C.Structure):
    _fields_ = [("f", C.c_void_p), ("c", C.c_void_p), ("wf", C.c_void_p), ("wc", C.c_void_p)]


class KSet(C.Structure):
    _fields_ = [("k", C.c_void_p * 7), ("coef", C.c_float * 7), ("nk", C.c_int)]


class Dopri5Advect(C.Structure):
    _fields_ = [("y", C.c_void_p * 2), ("f", C.c_void_p * 2), ("k", C.c_void_p * 5),
                ("Vx", C.c_void_p), ("Vy", C.c_void_p), ("Vz", C.c_void_p),
                ("sx", C.c_int), ("sy", C.c_int), ("sz", C.c_int), ("neumann_bc", C.c_int), ("is_f64", C.c_int),
                ("atol", C.c_double), ("rtol", C.c_double), ("tol_min_dt", C.c_double), ("dt_max", C.c_double),
                ("safety", C.c_double), ("ifactor", C.c_double), ("dfactor", C.c_double),
                ("t_out", C.c_void_p), ("nt", C.c_int), ("sol", C.c_void_p), ("state", C.c_void_p),
                ("workspace", C.c_void_p)]


GATHER_MAX_JOBS = 12          # BFM_GATHER_MAX_JOBS of include/brainfm_hip.h


class GatherJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("out", C.c_void_p), ("mean", C.c_float), ("scale", C.c_float), ("pre", C.c_int),
                ("default_max", C.c_int), ("post_div", C.c_float), ("clamp", C.c_int), ("clamp_lo", C.c_float),
                ("clamp_hi", C.c_float), ("sign", C.c_float), ("want_minmax", C.c_int)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_L = C.c_int64
_Z = C.c_size_t
_UP = C.POINTER(Upsample)

# name -> (restype, argtypes); every symbol declared in include/brainfm_hip.h
SIGNATURES = {
    "bfm_version": (C.c_char_p, []),
    "bfm_gn_stats_workspace": (_Z, [_I, _I, _I, _I, _I, _UP]),
    "bfm_gn_stats": (_I, [_P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _I, _F, _P, _P, _P, _P, _Z, _P]),
    "bfm_pack_conv_weights_direct_bytes": (_Z, [_I, _I]),
    "bfm_pack_conv_weights_direct": (_I, [_P, _I, _I, _P, _P]),
    "bfm_pack_conv_weights_mfma_bytes": (_Z, [_I, _I]),
    "bfm_pack_conv_weights_mfma": (_I, [_P, _I, _I, _F, _P, C.POINTER(_I), _P]),
    "bfm_pack_conv_weights_mfma16_bytes": (_Z, [_I, _I]),
    "bfm_pack_conv_weights_mfma16": (_I, [_P, _I, _I, _F, _P, C.POINTER(_I), _P]),
    "bfm_conv3x3x3_direct": (_I, [_P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _P, _I, _F, _P, _P]),
    "bfm_pack_conv_weights_upfold_bytes": (C.c_size_t, [_I, _I, _I]),
    "bfm_pack_conv_weights_upfold": (_I, [_P, _I, _I, _I, _F, _I, _P, _P, _P]),
    "bfm_conv3x3x3_upfold": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P]),
    "bfm_conv3x3x3_upfold_workspace": (_Z, [_I, _I, _I, _I, _I]),
    "bfm_conv3x3x3_upfold_ex": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _Z, _P]),
    "bfm_conv3x3x3_upfold_batch_workspace": (_Z, [_I, _I, _I, _I, _I, _I]),
    "bfm_conv3x3x3_upfold_batch": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _Z, _I, _P]),
    "bfm_moment_rows_bytes": (_Z, [_I, _I]),
    "bfm_conv3x3x3_mfma_rows": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I)]),
    "bfm_conv3x3x3_mfma_ex": (_I, [_P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _P, _I, _P, _I, _I, _F, _I,
                                   C.POINTER(_I), _P, _P, _Z, _P, _P]),
    "bfm_conv3x3x3_mfma_batch_workspace": (_Z, [_I, _I, _I, _I, _I, _I, _I]),
    "bfm_conv3x3x3_mfma_batch": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _UP, _P, _P, _P, _I, _P, _I, _I, _F, _I,
                                      C.POINTER(_I), _P, _P, _Z, _P, _I, _P]),
    "bfm_gn_stats_batch_workspace": (_Z, [_I, _I, _I, _I, _I, _I, _UP]),
    "bfm_gn_stats_batch": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _UP, _P, _P, _I, _F, _P, _P, _P, _P, _Z, _P]),
    "bfm_gn_stats_rows_batch": (_I, [_P, _I, _I, _P, _I, _I, C.c_double, _L, _I, _P, _P, _I, _F, _P, _P, _P, _P]),
    "bfm_conv3x3x3_stem_rows": (_I, [_I, _I, _I]),
    "bfm_conv3x3x3_stem_ex": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P, _P]),
    "bfm_gn_stats_rows_workspace": (_Z, [_I, _I, _I, _I]),
    "bfm_gn_stats_rows": (_I, [_P, _I, _I, _P, _I, _I, C.c_double, _L, _P, _P, _I, _F, _P, _P, _P, _P, _Z, _P, _P]),
    "bfm_gn_stats_rows_sliced": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, C.c_double, _L, _P, _P, _I, _F, _P, _P, _P, _P, _Z,
                                      _P, _P]),
    "bfm_gn_stats_rows_train": (_I, [_P, _I, _I, _P, _I, _I, C.c_double, _L, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P, _Z, _P, _P]),
    "bfm_crop3d": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "bfm_pack_conv_weights_wino4_bytes": (_Z, [_I, _I, _I]),
    "bfm_pack_conv_weights_wino4": (_I, [_P, _I, _I, _F, _I, _P, C.POINTER(_I), _P]),
    "bfm_conv3x3x3_wino4_rows": (_I, [_I, _I, _I, _I]),
    "bfm_conv3x3x3_wino4_box": (_I, [_I, _I, _I, _I, _P]),
    "bfm_conv3x3x3_wino4_masked_workspace": (_Z, [_I, _I, _I, _I]),
    "bfm_conv3x3x3_wino4_uniform_scratch": (_Z, [_I]),
    "bfm_conv3x3x3_wino4_uniform": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P]),
    "bfm_conv3x3x3_wino4": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P]),
    "bfm_conv3x3x3_wino4_batch": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _I, _P]),
    "bfm_conv3x3x3_wino4_masked": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "bfm_permute_flip3d": (_I, [_P, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _P, _P]),
    "bfm_bbox_nonzero": (_I, [_P, _I, _I, _I, _F, _P, _P]),
    "bfm_mean_lastdim": (_I, [_P, _L, _I, _P, _P]),
    "bfm_pack_conv_weights_wino_bytes": (_Z, [_I, _I, _I]),
    "bfm_pack_conv_weights_wino": (_I, [_P, _I, _I, _F, _I, _P, C.POINTER(_I), _P]),
    "bfm_conv3x3x3_wino": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P]),
    "bfm_bspline3_prefilter_axis": (_I, [_P, _I, _I, _I, _I, _I, _F, _F, _P, _F, _F, _F, _P]),
    "bfm_bspline3_resample_axis": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P, _P]),
    "bfm_conv3x3x3_wino_rows": (_I, [_I, _I, _I, _I]),
    "bfm_conv3x3x3_wino_ex": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P]),
    "bfm_conv3x3x3_wino_box": (_I, [_I, _I, _I, _I, _P]),
    "bfm_uniform_boxes": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "bfm_uniform_boxes_level": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "bfm_uniform_boxes_bytes": (_Z, [_I, _I, _I, _I]),
    "bfm_conv3x3x3_wino_uniform_scratch": (_Z, [_I]),
    "bfm_conv3x3x3_wino_uniform": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P]),
    "bfm_conv3x3x3_wino_pool_ok": (_I, [_I, _I, _I, _I]),
    "bfm_conv3x3x3_wino_pool": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P]),
    "bfm_conv3x3x3_wino_uniform_pool": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P, _P,
                                             _P]),
    "bfm_conv3x3x3_wino_masked_workspace": (_Z, [_I, _I, _I, _I]),
    "bfm_conv3x3x3_wino_masked": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "bfm_maxpool2_rows": (_I, [_I, _I, _I, _I]),
    "bfm_maxpool2_ex": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "bfm_maxpool2_batch_rows": (_I, [_I, _I, _I, _I]),
    "bfm_maxpool2_batch": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "bfm_grid_push3d_linear": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, C.POINTER(_I), _I, _P, _P]),
    "bfm_grid_grad3d_linear": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, C.POINTER(_I), _I, _P, _P]),
    "bfm_gn_stats_train": (_I, [_P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bfm_lrelu_bwd": (_I, [_P, _P, _L, _F, _P, _P]),
    "bfm_lrelu_bwd_ex": (_I, [_P, _P, _L, _F, _P, _P, _P]),
    "bfm_conv3x3x3_wgrad_workspace": (_Z, [_I, _I, _I, _I, _I]),
    "bfm_conv3x3x3_wgrad": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _P, _P, _Z, _P]),
    "bfm_conv3x3x3_wgrad_ex": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _P, _P, _I, _I, _P, _P, _Z, _P]),
    "bfm_gn_bwd_workspace": (_Z, [_I, _I, _I, _I]),
    "bfm_gn_bwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "bfm_maxpool2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "bfm_transpose_mirror_weights": (_I, [_P, _I, _I, _I, _P, _P]),
    "bfm_loss_workspace": (_Z, [_I]),
    "bfm_loss_l1": (_I, [_P, _I, _I, _P, _P, _P, _L, _F, _I, _F, _P, _P, _P, _Z, _P]),
    "bfm_loss_l1_multi_workspace": (_Z, []),
    "bfm_loss_l1_multi": (_I, [_P, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bfm_loss_grad_l1_multi": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "bfm_loss_grad_l1": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "bfm_loss_seg": (_I, [_P, _I, _I, _I, _P, _P, _P, _L, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "bfm_loss_pathol_workspace": (_Z, []),
    "bfm_loss_pathol": (_I, [_P, _L, _L, _P, _L, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "bfm_head_bwd_workspace": (_Z, [_I, _I, _L]),
    "bfm_head_bwd": (_I, [_P, _P, _P, _I, _I, _L, _P, _P, _P, _P, _Z, _P]),
    "bfm_tail_raw_rows": (_I, [_P, _L, C.POINTER(TailDesc), _P, _P, _L, _P]),
    "bfm_loss_l1_multi_rows": (_I, [_P, _L, _I, _L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "bfm_loss_grad_l1_multi_rows": (_I, [_P, _L, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "bfm_loss_seg_rows": (_I, [_P, _L, _I, _I, _I, _P, _P, _P, _L, _F, _F, _P, _P, _P, _P, _Z, _P]),
    "bfm_head_bwd_rows": (_I, [_P, _L, _P, _P, _I, _I, _L, _P, _P, _P, _P, _Z, _P]),
    "bfm_normalize_bwd": (_I, [_P, _P, _I, _L, _F, _P, _P]),
    "bfm_adamw_step": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "bfm_grad_sumsq": (_I, [_P, _L, _P, _P, _P, _Z, _P]),
    "bfm_grad_sumsq_multi": (_I, [_P, _I, _P, _I, _L, _P, _P, _P, _Z, _P]),
    "bfm_adamw_step_multi": (_I, [_P, _I, _P, _I, _L, _F, _F, _F, _F, _F, _P]),
    "bfm_conv3x3x3_stem": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P, _P]),
    "bfm_conv3x3x3_mfma_workspace": (_Z, [_I, _I, _I, _I, _I, _I]),
    "bfm_conv3x3x3_mfma_plan": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I)]),
    "bfm_conv3x3x3_mfma": (_I, [_P, _I, _P, _I, _I, _I, _I, _UP, _P, _P, _P, _I, _P, _I, _I, _F, _I,
                                C.POINTER(_I), _P, _P, _Z, _P]),
    "bfm_maxpool2": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "bfm_tail_heads": (_I, [_P, _P, _L, C.POINTER(TailDesc), _P, _P, _P, _P, _P, _P]),
    "bfm_tail_heads_rows": (_I, [_P, _P, _L, C.POINTER(TailDesc), _P, _P, _L, _P, _P, _I, _P]),
    "bfm_ew_unary": (_I, [_I, _P, _L, _P, _L, _L, _F, _F, _P]),
    "bfm_ew_binary": (_I, [_I, _P, _L, _P, _L, _P, _L, _L, _F, _P]),
    "bfm_absmax_f32": (_I, [_P, _L, _L, _L, _P, _P]),
    "bfm_softmax_cl": (_I, [_P, _L, _I, _P, _L, _L, _P]),
    "bfm_argmax_lut_cl": (_I, [_P, _L, _I, _P, _P, _L, _P]),
    "bfm_pathology_encode": (_I, [_P, _P, _P, _P, _F, _F, _F, _F, _L, _P, _P]),
    "bfm_fake_cortical": (_I, [_P, _L, _I, _P, _L, _P]),
    "bfm_interp3d_linear": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _L, _F, _P, _P]),
    "bfm_interp3d_nearest": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P]),
    "bfm_deformed_atlas": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(_F), _L, _P, _P]),
    "bfm_deformed_atlas_tile": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(_F), _L, _P, _P]),
    "bfm_zoom_linear": (_I, [_P, _I, _I, _I, _I, C.POINTER(ZoomAxis), _I, _I, _I, _P, _P]),
    "bfm_conv1d_axis": (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "bfm_grid_pull3d_linear": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _I, C.POINTER(_I), _I, _P, _P]),
    "bfm_deform_grid_workspace": (_Z, [_I, _I, _I]),
    "bfm_deform_grid": (_I, [_P, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), C.POINTER(_I), _P, _P, _P, _P, _P, _Z, _P]),
    "bfm_label_gauss": (_I, [_P, _P, _P, _P, _L, _I, _P, _P]),
    "bfm_label_class_stats": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "bfm_onehot_lut": (_I, [_P, _P, _I, _I, _L, _P, _P]),
    "bfm_perlin3d": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "bfm_radix_hist_f64": (_I, [_P, _L, C.c_uint64, _I, _P, _P]),
    "bfm_threshold_mask_f64": (_I, [_P, _L, C.c_double, _P, _P, _P]),
    "bfm_curl3d": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "bfm_advect_upwind_rhs": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "bfm_rk_combine": (_I, [_P, _I, C.POINTER(KSet), _P, _L, _P]),
    "bfm_reduce_workspace": (_Z, []),
    "bfm_rk_error_sumsq": (_I, [C.POINTER(KSet), _P, _P, _I, C.c_double, C.c_double, _L, _P, _P, _Z, _P]),
    "bfm_scaled_sumsq": (_I, [_P, _P, _I, _P, _I, C.c_double, C.c_double, _L, _P, _P, _Z, _P]),
    "bfm_dopri5_dense_eval": (_I, [_P, _P, _I, C.POINTER(KSet), C.c_double, C.c_double, _P, _L, _P]),
    "bfm_reduce_f32": (_I, [_I, _P, _P, _L, _P, _P, _Z, _P]),
    "bfm_reduce_f64": (_I, [_I, _P, _P, _L, _P, _P, _Z, _P]),
    "bfm_dopri5_advect_state_bytes": (_Z, []),
    "bfm_dopri5_advect_workspace": (_Z, [_I, _I, _I]),
    "bfm_dopri5_advect_init": (_I, [C.POINTER(Dopri5Advect), C.c_double, C.c_double, _P]),
    "bfm_dopri5_advect_steps": (_I, [C.POINTER(Dopri5Advect), _I, _P]),
    "bfm_randn_philox": (_I, [_P, _L, C.c_uint64, C.c_uint64, _F, _P]),
    "bfm_deform_zoom_workspace": (_Z, []),
    "bfm_deform_zoom_minmax": (_I, [_P, _I, _I, _I, C.POINTER(ZoomAxis), _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F),
                                    C.POINTER(_I), _P, _P, _Z, _P]),
    "bfm_deform_zoom_write": (_I, [_P, _I, _I, _I, C.POINTER(ZoomAxis), _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F),
                                   C.POINTER(_I), C.POINTER(_F), _P, _P, _P, _P, _P]),
    "bfm_gather_targets_workspace": (_Z, []),
    "bfm_gather_targets": (_I, [C.POINTER(GatherJob), _I, _I, _I, _I, C.POINTER(_I), _P, _P, _P, _I, _I, _I, _I, _P, _P,
                                _Z, _P]),
    "bfm_minmax_normalise": (_I, [_P, _L, _P, _P]),
    "bfm_gather_onehot": (_I, [_P, _I, _I, _I, C.POINTER(_I), _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "bfm_gather_onehot_rows": (_I, [_P, _I, _I, _I, C.POINTER(_I), _P, _P, _P, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "bfm_percentile_workspace": (_Z, []),
    "bfm_percentile_f64": (_I, [_P, _L, _L, _I, C.c_double, _P, _P, _Z, _P]),
    "bfm_shape_workspace": (_Z, []),
    "bfm_shape_threshold_f64": (_I, [_P, _L, _P, _P, _P, _P, _P, _Z, _P]),
    "bfm_shape_binarize": (_I, [_P, _I, _L, _P, C.c_double, _P, _P, _P, _Z, _P]),
    "bfm_pathology_mask": (_I, [_P, _P, _I, _P, _L, _P, _P, _Z, _P]),
    "bfm_pathology_encode_workspace": (_Z, []),
    "bfm_pathology_encode_dev": (_I, [_P, _P, _P, _I, _P, C.POINTER(_F), _I, _P, _L, _P, _P, _P, _Z, _P]),
    "bfm_interp3d_linear_axes": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P]),
    "bfm_sample_finalize": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P, _P]),
    "bfm_ew_dev": (_I, [_I, _P, _L, _P, _F, _P, _P]),
    "bfm_stitch_accumulate": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bfm_stitch_accumulate_multi": (_I, [_P, _L, _P, _I, _P, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "bfm_mask_tile": (_I, [_P, _P, _P, _L, _P, _P]),
    "bfm_tile_count_add": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "bfm_divide_by_count": (_I, [_P, _P, _L, _P]),
    "bfm_pack_tile_multi": (_I, [_P, _L, _P, _I, _P, _P, _L, _P, _P]),
    "bfm_divide_by_count_multi": (_I, [_P, _P, _L, _I, _P]),
    "bfm_stitch_gather_multi": (_I, [_P, _I, _I, _P, _I, _I, _I, _P]),
    "bfm_tile_mask_blocks": (_I, [_L]),
    "bfm_tile_mask_index": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _P, _P]),
    "bfm_pack_tile_compact": (_I, [_P, _L, _P, _I, _P, _P, _L, _P, _L, _P, _P]),
    "bfm_stitch_gather_compact": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P]),
}

_lib = None


def register(sigs):
    """Let sibling modules (synthesis kernels) add their symbols before load()."""
    SIGNATURES.update(sigs)
    if _lib is not None:
        _bind(_lib, sigs)


def _bind(lib, sigs):
    for name, (res, args) in sigs.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise BfmError("libbrainfm_hip.so does not export %s -- rebuild it" % name) from e
        fn.restype = res
        fn.argtypes = args


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BfmError("HIP extension missing: %s (run `python -m brainfm_amd.build`); "
                           "there is no CPU fallback in the product path" % LIB_PATH)
        import torch  # noqa: F401 -- first: torch ships its own libamdhip64; the extension must bind to THAT runtime (the one
        #                          whose streams and pointers it is handed), not to a second copy from /opt/rocm
        lib = C.CDLL(LIB_PATH)
        _bind(lib, SIGNATURES)
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise BfmError("%s failed: %s (%d)" % (what, ERR.get(rc, "?"), rc))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    """The current stream of the CURRENT device: every launch goes there, so an entry point that was given a device
    must make it current first (on_device below) -- pointers of one GPU on another GPU's stream fault."""
    # torch._C._cuda_getCurrentRawStream / _cuda_getDevice: the two C calls behind torch.cuda.current_stream().cuda_stream
    # without its Python layers (9.5 us -> < 1 us per launch: a generator item makes 100 launches, a training iteration 1 500)
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(pick):
    """Decorator: run the function with the device ``pick(*args, **kwargs)`` returns made current (torch.cuda.device),
    so that stream_ptr(), torch.cuda.current_stream() and fresh allocations all belong to the GPU the operands live on
    even when the caller's current device is another one.  ``pick`` may return a tensor, a torch.device, an int or a
    'cuda:N' string; CPU / None leaves the current device alone (the callee raises its own 'no CPU fallback' error)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapped(*a, **k):
            import torch
            d = pick(*a, **k)
            if isinstance(d, torch.Tensor):
                d = d.device
            if isinstance(d, int):
                d = torch.device("cuda", d)
            if isinstance(d, str):
                d = torch.device(d)
            if d is None or d.type != "cuda" or d.index is None or d.index == torch.cuda.current_device():
                return fn(*a, **k)
            with torch.cuda.device(d):
                return fn(*a, **k)
        return wrapped
    return deco
