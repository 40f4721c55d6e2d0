"""ctypes binding of libgeomae_hip.so (the C ABI declared in include/geomae_hip.h).

The HIP library IS the product: there is no CPU or eager-PyTorch fallback.  Importing
`geomae_amd` works without it (config / registry / host logic are importable on CPU), but any
compute entry point raises GeomaeLibraryError when the shared object is missing.
"""
import ctypes
import os
from ctypes import POINTER, c_double, c_float, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgeomae_hip.so")


class GeomaeLibraryError(RuntimeError):
    pass


class GeomaeTargetConfig(ctypes.Structure):
    _fields_ = [("grid_size", c_int32 * 3), ("ratio_low", c_int32 * 3), ("ratio_med", c_int32 * 3),
                ("voxel_size_top", c_float * 3), ("voxel_size_med", c_float * 3),
                ("voxel_size_low", c_float * 3), ("coors_range", c_float * 6)]


class GeomaeWindowConfig(ctypes.Structure):
    _fields_ = [("window_shape", c_int32 * 2), ("shift", c_int32 * 2), ("bev_shape", c_int32 * 2)]


class GeomaeWindowBuildJob(ctypes.Structure):
    _fields_ = ([("coors", c_void_p), ("num_tokens", c_int32), ("shift_index", c_int32)]
                + [(n, c_void_p) for n in ("win_start", "win_tokens", "tok_win", "tok_pos", "num_windows", "bun_start",
                                           "num_bundles", "bun_tok", "pos_info", "fbun_tok", "num_fbundles", "fitems", "num_fitems")])


class GeomaeSstLayerWeights(ctypes.Structure):
    _fields_ = ([(n, c_void_p) for n in ("wqkv_p", "wqkT_p", "wvT_p", "wo_p", "woT_p", "w1_p", "w1T_p", "w2_p",
                                         "w2T_p", "bqkv", "bo", "b1", "b2", "ln1_w", "ln1_b", "ln2_w", "ln2_b")]
                + [("d_model", c_int32), ("d_ffn", c_int32), ("ln_eps", c_float), ("frag_p", c_void_p)])


class GeomaeSstStackLayout(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("win_start", "win_tokens", "tok_win", "tok_pos", "bun_start", "num_bundles")] + \
               [("max_bundles", c_int32), ("bun_tok", c_void_p), ("pos_info", c_void_p), ("fbun_tok", c_void_p),
                ("num_fbundles", c_void_p), ("fitems", c_void_p), ("num_fitems", c_void_p)]


class GeomaeVfeArgs(ctypes.Structure):
    _fields_ = [("feat_sorted", c_void_p), ("pid_sorted", c_void_p), ("seg_start", c_void_p), ("num_points", c_int64),
                ("max_pillars", c_int32), ("w0", c_void_p), ("w1", c_void_p), ("scale0", c_void_p), ("shift0", c_void_p),
                ("scale1", c_void_p), ("shift1", c_void_p), ("moments", c_void_p), ("dw0_acc", c_void_p), ("pillar_ties", c_void_p),
                ("layer1_bf16", c_int32)]


class GeomaeBnFold(ctypes.Structure):
    _fields_ = [("count", c_double), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("momentum", c_float),
                ("running_mean", c_void_p), ("running_var", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("invstd", c_void_p), ("moments", c_void_p), ("num_batches_tracked", c_void_p)]


class GeomaeSweepInfo(ctypes.Structure):
    _fields_ = [("rot", c_double * 9), ("trans", c_double * 3), ("dt", c_float), ("frame", c_int32),
                ("remove_close", c_int32), ("has_transform", c_int32)]


class GeomaeFrameAug(ctypes.Structure):
    _fields_ = [("rot_cos", c_float), ("rot_sin", c_float), ("scale", c_float), ("trans", c_float * 3),
                ("flip_horizontal", c_int32), ("flip_vertical", c_int32), ("shuffle_seed_lo", ctypes.c_uint32),
                ("shuffle_seed_hi", ctypes.c_uint32)]


class GeomaeBnState(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("scale0", "shift0", "mean0", "invstd0", "scale1", "shift1", "mean1", "invstd1")]


class GeomaeHeadGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("reg_low_w", "reg_low_b", "cls_low_w", "cls_low_b", "reg_med_w", "reg_med_b",
                                        "cls_med_w", "cls_med_b", "reg_top_w", "reg_top_b", "nor_top_w", "nor_top_b")]


class GeomaeSstLayerGrads(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2", "ln1_w", "ln1_b",
                                        "ln2_w", "ln2_b")]


class GeomaePretrainConfig(ctypes.Structure):
    _fields_ = [("batch_size", c_int32), ("num_features", c_int32), ("targets", GeomaeTargetConfig),
                ("window", GeomaeWindowConfig), ("num_heads", c_int32), ("encoder_layers", c_int32),
                ("decoder_layers", c_int32), ("keep_fraction", c_double), ("mask_seed", c_uint64),
                ("loss_weights", c_float * 6), ("vfe_voxel_size", c_float * 3), ("vfe_center_offset", c_float * 3),
                ("bn_eps", c_float), ("bn_momentum", c_float), ("beta1", c_float), ("beta2", c_float),
                ("adam_eps", c_float), ("weight_decay", c_float), ("max_grad_norm", c_float), ("world_size", c_int32), ("sync_bn", c_int32),
                ("exchange_always", c_int32), ("vfe_bf16", c_int32)]


class GeomaePretrainModel(ctypes.Structure):
    _fields_ = ([("layers", c_void_p), ("layer_grads", c_void_p), ("head_grads", GeomaeHeadGrads),
                 ("head_w_packed", c_void_p), ("head_bias", c_void_p), ("pos_table", c_void_p), ("mask_token", c_void_p),
                 ("mask_token_grad", c_void_p), ("pack_desc", c_void_p), ("num_pack_desc", c_int32),
                 ("pack_max_elems", c_int64), ("packed", c_void_p), ("pack_aux", c_void_p),
                 ("vfe_w0", c_void_p), ("vfe_w1", c_void_p), ("vfe_dw0", c_void_p), ("vfe_dw1", c_void_p),
                 ("bn_gamma", c_void_p * 2), ("bn_beta", c_void_p * 2), ("bn_dgamma", c_void_p * 2),
                 ("bn_dbeta", c_void_p * 2), ("bn_running_mean", c_void_p * 2), ("bn_running_var", c_void_p * 2),
                 ("bn_num_batches", c_void_p * 2), ("params", c_void_p), ("grads", c_void_p), ("exp_avg", c_void_p),
                 ("exp_avg_sq", c_void_p), ("num_params", c_int64), ("no_decay_prefix", c_int64),
                 ("no_decay2_start", c_int64), ("no_decay2_count", c_int64), ("bn_sync_moments0", c_void_p),
                 ("bn_sync_moments1", c_void_p), ("bn_sync_bsums1", c_void_p), ("bn_sync_bsums0", c_void_p),
                 ("bn_sync_feat_moments", c_void_p)])


class GeomaeTuning(ctypes.Structure):
    """include/geomae_hip.h GeomaeTuning: the one surface of kernel-form / schedule switches."""
    _fields_ = [(n, c_int32) for n in (
        "size", "fused_layers", "fused_max_tokens", "fused_bwd", "ws_layers", "ws_bwd", "ws_bundle_cap", "ws_max_workgroups",
        "bundle_cap", "saved_f32", "x_from_xhat", "y_from_xhat", "pair_kernels", "attn_heads", "dw_layer_form", "dw_chunks",
        "dw_budget_mid", "dw_split_reduce", "dw_defer_all", "dec_dw_every", "dec_mid_budget", "enc_dw_defer", "zero_late_aux",
        "fused_skip_big", "heads_joint", "fwd_item_cap")] + [("reserved", c_int32 * 7)]


# environment variable -> GeomaeTuning field, read ONCE when the library is loaded (the library itself reads no environment)
TUNING_ENV = {
    "GEOMAE_FUSED_LAYERS": "fused_layers", "GEOMAE_FUSED_MAX_TOKENS": "fused_max_tokens", "GEOMAE_FUSED_BWD": "fused_bwd",
    "GEOMAE_WS_LAYERS": "ws_layers", "GEOMAE_WS_BWD": "ws_bwd", "GEOMAE_WS_BUNDLE_CAP": "ws_bundle_cap",
    "GEOMAE_WS_MAX_WORKGROUPS": "ws_max_workgroups", "GEOMAE_BUNDLE_CAP": "bundle_cap", "GEOMAE_SAVED_F32": "saved_f32",
    "GEOMAE_X_FROM_XHAT": "x_from_xhat", "GEOMAE_Y_FROM_XHAT": "y_from_xhat", "GEOMAE_PAIR_KERNELS": "pair_kernels",
    "GEOMAE_ATTN_HEADS": "attn_heads", "GEOMAE_DW_LAYER_FORM": "dw_layer_form", "GEOMAE_DW_CHUNKS": "dw_chunks",
    "GEOMAE_DW_BUDGET_MID": "dw_budget_mid", "GEOMAE_DW_SPLIT_REDUCE": "dw_split_reduce", "GEOMAE_DW_DEFER_ALL": "dw_defer_all",
    "GEOMAE_DEC_DW_EVERY": "dec_dw_every", "GEOMAE_DEC_MID_BUDGET": "dec_mid_budget", "GEOMAE_ENC_DW_DEFER": "enc_dw_defer",
    "GEOMAE_ZERO_LATE_AUX": "zero_late_aux", "GEOMAE_FUSED_SKIP_BIG": "fused_skip_big", "GEOMAE_HEADS_JOINT": "heads_joint", "GEOMAE_FWD_ITEM_CAP": "fwd_item_cap",
}


def get_tuning():
    t = GeomaeTuning()
    check(load().geomae_get_tuning(ctypes.byref(t)), "geomae_get_tuning")
    return t


def set_tuning(**fields):
    """Change fields of the process-wide GeomaeTuning (between steps); returns the PREVIOUS values of the fields set."""
    t = get_tuning()
    old = {k: getattr(t, k) for k in fields}
    for k, v in fields.items():
        if k not in dict(GeomaeTuning._fields_) or k in ("size", "reserved"):
            raise KeyError(f"GeomaeTuning has no field {k!r}")
        setattr(t, k, int(v))
    check(load().geomae_set_tuning(ctypes.byref(t)), "geomae_set_tuning")
    return old


def _apply_tuning_env(lib):
    t = GeomaeTuning()
    if lib.geomae_get_tuning(ctypes.byref(t)) != 0:
        raise GeomaeLibraryError("geomae_get_tuning failed")
    touched = False
    for env, field in TUNING_ENV.items():
        v = os.environ.get(env)
        if v is not None and v != "":
            setattr(t, field, int(v))
            touched = True
    if touched and lib.geomae_set_tuning(ctypes.byref(t)) != 0:
        raise GeomaeLibraryError("geomae_set_tuning failed: " + lib.geomae_last_error().decode("utf-8", "replace"))


PRETRAIN_HOOK = ctypes.CFUNCTYPE(None, c_void_p, c_int32, c_void_p)

F3 = POINTER(c_float)
P = c_void_p
# name -> (restype, argtypes).  Every symbol declared in include/geomae_hip.h is listed here;
# tests/test_abi.py checks the header, this table and the built library against each other.
SIGNATURES = {
    "geomae_last_error": (ctypes.c_char_p, []),
    "geomae_abi_version": (c_int32, []),
    "geomae_grid_size": (ctypes.c_int, [F3, F3, POINTER(c_int32)]),
    "geomae_dynamic_voxelize": (ctypes.c_int, [P, c_int64, c_int32, F3, F3, P, P]),
    "geomae_voxelize_batch3": (ctypes.c_int, [P, c_int64, c_int32, P, c_int32, F3, F3, F3, F3, P, P, P, P]),
    "geomae_voxelize_frames3": (ctypes.c_int, [POINTER(c_void_p), POINTER(c_int64), c_int32, c_int32, F3, F3, F3, F3, P, P, P, P, P,
                                               P, c_int64, P, c_int64, P]),
    "geomae_pillar_segment_ex": (ctypes.c_int, [P, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, P, P, P, P, P,
                                                P, P, P, c_int64, P, c_int32, P]),
    "geomae_pillar_segment_scan_state_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "geomae_pillar_segment_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32, c_int32, c_int32]),
    "geomae_pillar_segment": (ctypes.c_int, [P, c_int64, c_int32, c_int32, c_int32, c_int32, P, P, P, P, P, P, P,
                                             P, c_int64, P]),
    "geomae_pillar_segment_nd": (ctypes.c_int, [P, c_int32, c_int64, c_int32, c_int32, c_int32, c_int32, P, P, P, P, P,
                                                P, P, P, c_int64, P]),
    "geomae_segment_mean_xyz": (ctypes.c_int, [P, c_int32, c_int64, P, P, P, c_int32, P, P, P]),
    "geomae_segment_mean_xyz_sorted": (ctypes.c_int, [P, c_int32, P, P, P, c_int32, P, P]),
    "geomae_vfe_prepare": (ctypes.c_int, [P, c_int32, c_int64, P, P, P, P, F3, F3, P, P, P]),
    "geomae_vfe_moments_workspace_bytes": (c_int64, []),
    "geomae_vfe_prepare_moments": (ctypes.c_int, [P, c_int32, c_int64, P, P, P, P, F3, F3, P, P, P, P, P]),
    "geomae_dynamic_point_to_voxel_workspace_bytes": (c_int64, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "geomae_dynamic_point_to_voxel_forward": (ctypes.c_int, [P, P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                                             c_int32, c_int32, c_int32, P, P, P, P, P, P, c_int64, P]),
    "geomae_dynamic_point_to_voxel_backward": (ctypes.c_int, [P, P, P, P, P, P, c_int64, c_int32, c_int32, c_int32, P, P]),
    "geomae_hard_voxelize_workspace_bytes": (c_int64, [c_int64, F3, F3]),
    "geomae_hard_voxelize": (ctypes.c_int, [P, c_int64, c_int32, F3, F3, c_int32, c_int32, P, P, P, P, P, c_int64, P]),
    "geomae_points_pipeline_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "geomae_points_pipeline": (ctypes.c_int, [P, c_int64, c_int32, P, P, c_int32, P, P, c_int32, F3, c_float, P, P, P,
                                              c_int64, P]),
    "geomae_window_drop_workspace_bytes": (c_int64, [c_int32, c_int32, POINTER(GeomaeWindowConfig)]),
    "geomae_window_drop": (ctypes.c_int, [P, c_int32, c_int32, POINTER(GeomaeWindowConfig), c_int32, c_int32,
                                          POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), P, P, P, c_int64, P]),
    "geomae_recover_bev_forward": (ctypes.c_int, [P, P, c_int64, c_int32, c_int32, c_int32, c_int32, P, P]),
    "geomae_recover_bev_backward": (ctypes.c_int, [P, P, c_int64, c_int32, c_int32, c_int32, c_int32, P, P]),
    "geomae_grad_sumsq": (ctypes.c_int, [P, c_int64, P, P]),
    "geomae_adamw_step": (ctypes.c_int, [P, P, P, P, c_int64, c_int64, c_float, c_float, c_float, c_float, c_float,
                                         c_int64, c_float, P, c_float, c_int32, P, c_int64, c_int64, P, P]),
    "geomae_bn_finalize": (ctypes.c_int, [P, c_double, P, c_int32, P, P, c_float, c_float, c_int32, P, P, P, P, P, P, P, P]),
    "geomae_vfe_stats0": (ctypes.c_int, [POINTER(GeomaeVfeArgs), P, P]),
    "geomae_vfe_layer0": (ctypes.c_int, [POINTER(GeomaeVfeArgs), P, P, P]),
    "geomae_vfe_layer1": (ctypes.c_int, [POINTER(GeomaeVfeArgs), P, P, P]),
    "geomae_vfe_layer0_bn": (ctypes.c_int, [POINTER(GeomaeVfeArgs), POINTER(GeomaeBnFold), P, P, P]),
    "geomae_vfe_layer1_bn": (ctypes.c_int, [POINTER(GeomaeVfeArgs), POINTER(GeomaeBnFold), P, P, P, P]),
    "geomae_vfe_backward_stats": (ctypes.c_int, [POINTER(GeomaeVfeArgs), POINTER(GeomaeBnState), P, P, P, P, P]),
    "geomae_vfe_backward_layer1": (ctypes.c_int, [POINTER(GeomaeVfeArgs), POINTER(GeomaeBnState), P, P, P, P, c_float,
                                                  P, P, P, P, P, P, P, P, P]),
    "geomae_vfe_backward_layer0": (ctypes.c_int, [POINTER(GeomaeVfeArgs), POINTER(GeomaeBnState), P, P, c_float, c_int64,
                                                  P, P, P, P, P, P, P]),
    "geomae_segment_max_forward": (ctypes.c_int, [P, c_int32, P, P, P, c_int32, P, P, P]),
    "geomae_segment_max_backward": (ctypes.c_int, [P, P, P, c_int64, c_int32, P, P]),
    "geomae_random_mask": (ctypes.c_int, [P, c_int32, c_double, c_uint64, P, P, P, P, P]),
    "geomae_random_mask_windowed": (ctypes.c_int, [P, c_int32, c_double, c_uint64, P, POINTER(GeomaeWindowConfig), P, P, P, P, P]),
    "geomae_geometry_targets": (ctypes.c_int, [P, c_int32, P, P, P, c_int32, P, P, P, P, c_int32, P, P,
                                               POINTER(GeomaeTargetConfig), P, P, P, P, P, P, P, P, P, P, P, P, c_int32, P]),
    "geomae_window_build_workspace_bytes": (c_int64, [c_int32, c_int32, POINTER(GeomaeWindowConfig)]),
    "geomae_window_build": (ctypes.c_int, [P, c_int32, c_int32, POINTER(GeomaeWindowConfig), c_int32, P, P, P, P,
                                           P, P, P, P, c_int64, P]),
    "geomae_window_build_batch_workspace_bytes": (c_int64, [POINTER(c_int32), c_int32, c_int32,
                                                            POINTER(GeomaeWindowConfig)]),
    "geomae_window_build_batch": (ctypes.c_int, [POINTER(GeomaeWindowBuildJob), c_int32, c_int32,
                                                 POINTER(GeomaeWindowConfig), P, c_int64, P]),
    "geomae_window_attention_forward": (ctypes.c_int, [P, c_int32, c_int32, c_int32, P, P, P, P, P, c_int32,
                                                       c_int32, P, P, P, P, P]),
    "geomae_window_attention_backward": (ctypes.c_int, [P, P, P, P, c_int32, c_int32, c_int32, P, P, P, P, P,
                                                        c_int32, c_int32, P, P, P, P]),
    "geomae_pack_weights": (ctypes.c_int, [P, P, c_int32, c_int64, P, P, P]),
    "geomae_heads_loss": (ctypes.c_int, [P, P, c_int32, c_int32, P, P, P, P, P, P, P, P, P, F3, P, P, P, P, P, P, P]),
    "geomae_gather_token_coors": (ctypes.c_int, [P, c_int32, P, c_int32, P, P, P, P]),
    "geomae_gather_token_coors_zero": (ctypes.c_int, [P, c_int32, P, c_int32, P, P, P, P, c_int64, P]),
    "geomae_window_build_batch_table_bytes": (c_int64, [POINTER(c_int32), c_int32, c_int32, POINTER(GeomaeWindowConfig)]),
    "geomae_set_accumulators_prezeroed": (ctypes.c_int, [c_int32]),
    "geomae_get_tuning": (ctypes.c_int, [POINTER(GeomaeTuning)]),
    "geomae_set_tuning": (ctypes.c_int, [POINTER(GeomaeTuning)]),
    "geomae_heads_loss_accumulate": (ctypes.c_int, [P, P, c_int32, c_int32, P, P, P, P, P, P, P, P, P, F3, P, P, P, P, P, P, P]),
    "geomae_sst_set_pair_kernels": (None, [c_int32]),
    "geomae_heads_loss_split_accumulate": (ctypes.c_int, [P, P, c_int32, c_int32, P, P, P, P, P, P, P, P, P, F3, P, P, P, P, P, P, P, P]),
    "geomae_heads_loss_centroid_accumulate": (ctypes.c_int, [P, c_int32, c_int32, P, P, P, P, P, P, P, P, F3, P, P, P, P, P, P]),
    "geomae_heads_loss_density_accumulate": (ctypes.c_int, [P, c_int32, c_int32, P, P, P, F3, P, P, P, P, P]),
    "geomae_heads_weight_grad": (ctypes.c_int, [c_int32, P, P, P, POINTER(GeomaeHeadGrads), P]),
    "geomae_sst_qkv_forward": (ctypes.c_int, [P, P, P, POINTER(GeomaeSstLayerWeights), c_int32, P, P, P, P]),
    "geomae_sst_ffn_forward": (ctypes.c_int, [P, P, POINTER(GeomaeSstLayerWeights), c_int32, P, P, P, P, P, P]),
    "geomae_sst_ffn_qkv_forward": (ctypes.c_int, [P, P, POINTER(GeomaeSstLayerWeights), c_int32, P, P, P, P, P,
                                                  POINTER(GeomaeSstLayerWeights), P, P, P, P, P, P]),
    "geomae_sst_ffn_backward": (ctypes.c_int, [P, P, P, P, P, POINTER(GeomaeSstLayerWeights), c_int32, P, P, P, P,
                                               P, P, P, POINTER(GeomaeSstLayerGrads), P, P, POINTER(GeomaeSstLayerWeights), P]),
    "geomae_sst_qkv_backward": (ctypes.c_int, [P, P, POINTER(GeomaeSstLayerWeights), c_int32, P, P]),
    "geomae_sst_weight_grad": (ctypes.c_int, [c_int32, P, P, P, P, P, P, P, P, P, POINTER(GeomaeSstLayerGrads), P]),
    "geomae_window_bundle_cap": (c_int32, [c_int32, c_int32]),
    "geomae_sst_layer_forward": (ctypes.c_int, [P, c_int32, POINTER(GeomaeSstLayerWeights), POINTER(GeomaeSstStackLayout),
                                                c_int32, P, P, c_int32, P, P, P, P, P, P, P, P, P, P]),
    "geomae_sst_set_fused_layers": (None, [c_int32]),
    "geomae_sst_last_stack_forms": (ctypes.c_int, [POINTER(c_int32)]),
    "geomae_sst_set_big_bundle_layouts": (None, [c_int32]),
    "geomae_sst_fused_dropped_bundles": (ctypes.c_int, [POINTER(c_int64), c_int32]),
    "geomae_sst_stack_saved_bytes": (c_int64, [c_int32, c_int32, c_int32]),
    "geomae_sst_stack_scratch_bytes": (c_int64, [c_int32]),
    "geomae_sst_stack_scratch_bytes_layers": (c_int64, [c_int32, c_int32]),
    "geomae_sst_stack_forward": (ctypes.c_int, [P, c_int32, P, c_int32, P, P, c_int32, c_int32, P, c_int64, P, c_int32, P,
                                                P, P, P]),
    "geomae_sst_stack_backward": (ctypes.c_int, [P, P, c_int32, P, P, c_int32, P, P, c_int32, c_int32, P, P, c_int64, P,
                                                 P, c_int32, P, c_int32, c_int32, P, P]),
    "geomae_flush_weight_grad": (ctypes.c_int, [P]),
    "geomae_vfe_weight_grad1": (ctypes.c_int, [P, P, c_int64, P, P]),
    "geomae_vfe_weight_grad1_workspace_bytes": (c_int64, []),
    "geomae_vfe_weight_grad1_ws": (ctypes.c_int, [P, P, c_int64, P, P, c_int64, P]),
    "geomae_window_rank": (ctypes.c_int, [P, P, P, c_int64, P, P, P]),
    "geomae_rows_scatter": (ctypes.c_int, [P, P, c_int64, c_int32, P, P]),
    "geomae_rows_gather": (ctypes.c_int, [P, P, c_int64, c_int32, P, P]),
    "geomae_bn_param_grad_add": (ctypes.c_int, [P, c_int32, P, P, P]),
    "geomae_pretrain_workspace_bytes": (c_int64, [POINTER(GeomaePretrainConfig), c_int64, c_int32]),
    "geomae_pretrain_create": (c_void_p, [POINTER(GeomaePretrainConfig), POINTER(GeomaePretrainModel), P, c_int64, c_int64,
                                          c_int32, POINTER(c_void_p)]),
    "geomae_pretrain_destroy": (None, [c_void_p]),
    "geomae_pretrain_set_hook": (ctypes.c_int, [c_void_p, PRETRAIN_HOOK, c_void_p]),
    "geomae_pretrain_set_profiler": (ctypes.c_int, [c_void_p, c_void_p]),
    "geomae_pretrain_set_phase_timing": (ctypes.c_int, [c_void_p, c_int32]),
    "geomae_pretrain_phase_times": (c_int32, [c_void_p, POINTER(c_float), c_int32]),
    "geomae_pretrain_invalidate_packed": (ctypes.c_int, [c_void_p]),
    "geomae_pretrain_submit": (ctypes.c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), P]),
    "geomae_pretrain_submit_ex": (ctypes.c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_int32, P]),
    "geomae_pretrain_pending_slot": (c_int32, [c_void_p]),
    "geomae_pretrain_set_mask": (ctypes.c_int, [c_void_p, P, c_int32, P, c_int32, P]),
    "geomae_pretrain_set_mask_draws": (ctypes.c_int, [c_void_p, ctypes.c_uint64]),
    "geomae_pretrain_step": (ctypes.c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), c_float, c_float, c_int32, P]),
    "geomae_pretrain_optimizer": (ctypes.c_int, [c_void_p, c_float, c_float, P]),
    "geomae_pretrain_result_offset": (c_int64, [c_void_p, c_int32]),
    "geomae_pretrain_host_times": (ctypes.c_int, [c_void_p, POINTER(c_double)]),
    "geomae_pretrain_set_optimizer_steps": (ctypes.c_int, [c_void_p, c_int64]),
    "geomae_pretrain_last_sizes": (ctypes.c_int, [c_void_p, POINTER(c_int64)]),
    "geomae_pretrain_last_sizes_n": (ctypes.c_int, [c_void_p, POINTER(c_int64), c_int32]),
    "geomae_pretrain_step_forms": (ctypes.c_int, [c_void_p, POINTER(c_int32), c_int32]),
    "geomae_profiler_create": (c_void_p, [c_int32, c_int32]),
    "geomae_profiler_read": (c_int32, [c_void_p, POINTER(c_float), c_int32]),
    "geomae_profiler_destroy": (None, [c_void_p]),
}

_lib = None


def load(path=None):
    """dlopen the HIP library (fails loudly -- there is no fallback path)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # GEOMAE_LIB=<path>: another build of the same library (tools/build_timing.py variants in A/B runs); still no fallback
    path = path or os.environ.get("GEOMAE_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise GeomaeLibraryError(
            f"{path} not found: build it with `python -m geomae_amd.csrc.build` "
            "(hipcc --offload-arch=gfx950).  geomae_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = ABI mismatch, loud
        fn.restype = res
        fn.argtypes = args
    if lib.geomae_abi_version() != 1:
        raise GeomaeLibraryError(f"ABI version {lib.geomae_abi_version()} != 1")
    lib.geomae_last_error.restype = ctypes.c_char_p
    _apply_tuning_env(lib)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().geomae_last_error().decode("utf-8", "replace")
        raise GeomaeLibraryError(f"{what} failed ({rc}): {msg}")


def f3(values):
    return (c_float * len(values))(*[float(v) for v in values])
