"""ctypes binding of ``libcatre_hip.so`` (C ABI declared in ``include/catre_hip.h``).

PyTorch is only plumbing here: device memory (``tensor.data_ptr()``), the current HIP stream
and, elsewhere, ``torch.distributed``.  There is NO fallback: if the library is missing or a
tensor is not on a HIP device, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CATRE_HIP_LIB selects another build of the same ABI (the instrumented `make TRACE=1` library for profiles/trace_*.py)
LIB_PATH = os.environ.get("CATRE_HIP_LIB") or os.path.join(_HERE, "csrc", "libcatre_hip.so")

# state_dict keys in the order of `enum catre_param` (include/catre_hip.h)
PARAM_KEYS = (
    [f"pcl_net.stn.{l}.{p}" for l in ("conv1", "conv2", "conv3", "fc1", "fc2", "fc3") for p in ("weight", "bias")]
    + [f"pcl_net.{l}.{p}" for l in ("conv1", "conv2", "conv3", "conv4") for p in ("weight", "bias")]
    + [f"pcl_net.fstn.{l}.{p}" for l in ("conv1", "conv2", "conv3", "fc1", "fc2", "fc3") for p in ("weight", "bias")]
    + [
        f"rot_head.rot_head_{a}.{l}.{p}"
        for a in ("x", "y")
        for l in ("layers.0", "layers.1", "layers.3", "layers.4", "neck.0", "conv_p")
        for p in ("weight", "bias")
    ]
    + [
        f"ts_head.{l}.{p}"
        for l in ("linears.0", "linears.1", "linears.3", "linears.4", "fc_t", "fc_s")
        for p in ("weight", "bias")
    ]
)
CATRE_P_COUNT = len(PARAM_KEYS)
assert CATRE_P_COUNT == 68, CATRE_P_COUNT


class CatreOpts(ctypes.Structure):
    _fields_ = [
        ("feature_transform", ctypes.c_int32),
        ("with_kps_feature", ctypes.c_int32),
        ("with_init_scale", ctypes.c_int32),
        ("with_init_trans", ctypes.c_int32),
        ("delta_t_space_3d", ctypes.c_int32),
        ("delta_z_deepim", ctypes.c_int32),
        ("k_aware", ctypes.c_int32),
        ("scale_mul", ctypes.c_int32),
        ("scale_base_mean", ctypes.c_int32),
        ("is_allo", ctypes.c_int32),
        ("refine_scale", ctypes.c_int32),
        ("zero_center", ctypes.c_int32),
        ("delta_t_weight", ctypes.c_float),
        ("allo_eps", ctypes.c_float),
        ("ts_in_dim", ctypes.c_int32),
        ("rot_input_is_matrix", ctypes.c_int32),
        ("compute_dtype", ctypes.c_int32),
        ("rot_type", ctypes.c_int32),
    ]


DTYPE_F32, DTYPE_BF16, DTYPE_SPLIT = 0, 1, 2
ROWS_BF16 = 0x100   # CATRE_ROWS_BF16
PACK_F32_ENCODER, PACK_F32_HEADS, PACK_BF16, PACK_SPLIT, PACK_F32_TAILS, PACK_ALL = 1, 2, 4, 8, 16, 31
ROT_6D, ROT_QUAT, ROT_LOG_QUAT, ROT_LIE_VEC = 0, 1, 2, 3
ROT_DIMS = {ROT_6D: 6, ROT_QUAT: 4, ROT_LOG_QUAT: 3, ROT_LIE_VEC: 3}


def rot_type_id(rot_type):
    """'{ego,allo}_{rot6d,quat,log_quat,lie_vec}' -> CATRE_ROT_* (reference models/model_utils.py:11-40)."""
    for suffix, v in (("_rot6d", ROT_6D), ("_log_quat", ROT_LOG_QUAT), ("_quat", ROT_QUAT), ("_lie_vec", ROT_LIE_VEC)):
        if rot_type in ("ego" + suffix, "allo" + suffix):
            return v
    raise ValueError(f"Unknown rot_type: {rot_type}")  # model_utils.py:24


class CatreLossCfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("pm_on", "pm_sym", "pm_with_scale", "rot_on", "rot_l2", "yaxis_smooth",
                                             "trans_on", "trans_mse", "trans_split", "scale_on", "scale_mse")] + \
               [(n, ctypes.c_float) for n in ("pm_lw", "rot_lw", "trans_lw", "scale_lw")]


class CatrePoints(ctypes.Structure):
    _fields_ = [
        ("obs", ctypes.c_void_p), ("obs_sb", ctypes.c_int64), ("obs_sn", ctypes.c_int64), ("obs_sc", ctypes.c_int64),
        ("kps", ctypes.c_void_p), ("kps_sb", ctypes.c_int64), ("kps_sn", ctypes.c_int64), ("kps_sc", ctypes.c_int64),
        ("pose", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("apply_pose", ctypes.c_int32), ("zero_center", ctypes.c_int32),
    ]


_P = ctypes.c_void_p
_I = ctypes.c_int
_SZ = ctypes.c_size_t
_SIGS = {
    "catre_workspace_bytes": (_SZ, [_I, _I, _I]),
    "catre_packed_floats": (_SZ, [_I, _I, _I]),
    "catre_pack_weights": (_I, [_P, _I, _I, _I, _P, _SZ, _P]),
    "catre_pack_weights_sel": (_I, [_P, _I, _I, _I, _P, _SZ, _I, _P]),
    "catre_op_rows_compact": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "catre_op_maxlin_bwd_x_compact": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_x_compact_h": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_x_compact_cm": (_I, [_P, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_w_c": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "catre_op_stn_recompute": (_I, [_I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "catre_op_gather_rows": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "catre_op_scatter_rows": (_I, [_P, _I, _P, _P, _I, _I, _I, _P]),
    "catre_op_scatter_rows_merge": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "catre_op_gemm_rows_n": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "catre_op_gemm_tn_bias_n": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _SZ, _P, _I, _P]),
    "catre_op_gemm_rows_nr": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "catre_op_gemm_tn_bias_nr": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _SZ, _P, _I, _P]),
    "catre_train_stn3d_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "catre_train_stnkd_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "catre_train_trunk_fwd": (_I, [_P] * 12 + [_P, _SZ, _I, _I, _I, _I, _P]),
    "catre_pose_apply": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "catre_stn3d_pool": (_I, [_P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "catre_linear": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "catre_linear_t": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "catre_stnkd_pool": (_I, [_P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "catre_trunk": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "catre_ts_head": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _P]),
    "catre_rot_head": (_I, [_P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "catre_rot_head_dim": (_I, [_P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "catre_rot_to_mat": (_I, [_P, _I, _P, _I, _P]),
    "catre_rot_to_mat_bwd": (_I, [_P, _I, _P, _P, _I, _P]),
    "catre_pose_update": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "catre_refine_iter": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _P]),
    "catre_refine_k": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "catre_form_switch": (_I, [_I, _I]),
    "catre_refine_k_from": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _I, _I, _P]),
    "catre_colmax": (_I, [_P, _P, _I, _I, _I, _P]),
    # training ops (include/catre_hip.h "training ops")
    "catre_op_pack": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "catre_op_gemm_rows": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "catre_op_linear_maxpool_ws_bytes": (_SZ, [_I, _I]),
    "catre_op_linear_maxpool": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_pack_bf16": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "catre_op_gemm_rows_bf16": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "catre_op_linear_maxpool_bf16": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_gemm_rows_cloudbias": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "catre_op_gemm_rows_gn": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "catre_op_gnp_gelu_fwd_pre": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "catre_op_pack_split": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "catre_op_gemm_rows_split": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "catre_op_linear_maxpool_split": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_gemm_tn_ws_bytes": (_SZ, [_I, _I, _I]),
    "catre_op_gemm_tn": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_gemm_rows_m": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "catre_op_gemm_tn_bias_m": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_gemm_tn_bias_lp": (_I, [_P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _SZ, _I, _P]),
    "catre_op_gemm_tn_bias_ws_bytes": (_SZ, [_I, _I, _I]),
    "catre_op_fc_bwd": (_I, [_P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "catre_op_skinny_bwd": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_gemm_tn_bias": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _SZ, _P]),
    "catre_op_colsum": (_I, [_P, _I, _I, _I, _P, _I, _P, _SZ, _P]),
    "catre_op_reduce_splits": (_I, [_P, _P, _I, _I, _I, _P]),
    "catre_op_rowbias_add": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "catre_op_rowbias_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    "catre_op_maxpool_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "catre_op_maxpool_scatter": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_w": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_w_h": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_x": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _P]),
    "catre_op_maxlin_bwd_x_rows": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "catre_op_cloud_matmul": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "catre_op_cloud_matmul_bwd_t": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P]),
    "catre_op_relu_bwd": (_I, [_P, _P, _P, _SZ, _P]),
    "catre_op_sum_rows": (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "catre_op_gnp_gelu_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "catre_op_gnp_gelu_neck_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "catre_op_gnp_gelu_neck_bwd_ws_bytes": (_SZ, [_I, _I]),
    "catre_op_gnp_gelu_neck_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _P]),
    "catre_op_rot_l0_bwd_ws_bytes": (_SZ, [_I, _I, _I]),
    "catre_op_rot_l0_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _I, _P]),
    "catre_op_rot_l0_bwd_lp": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _I, _P]),
    "catre_op_rot_l0_bwd_sp": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _I, _P]),
    "catre_op_gnp_gelu_neck_fwd_s": (_I, [_P] * 10 + [_I, _I, _P]),
    "catre_op_rot_l1_bwd_s": (_I, [_P] * 14 + [_SZ, _I, _I, _P]),
    "catre_op_rot_l1_bwd_lp": (_I, [_P] * 14 + [_SZ, _I, _I, _P]),
    "catre_op_rot_l1_bwd_sp": (_I, [_P] * 14 + [_SZ, _I, _I, _P]),
    "catre_op_rot_l1_bwd_h": (_I, [_P] * 14 + [_SZ, _I, _I, _P]),
    "catre_op_rot_l0_bwd_h": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _I, _P]),
    "catre_op_gemm_rows_gn_h": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "catre_op_gnp_gelu_fwd_pre_h": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "catre_op_gnp_gelu_neck_fwd_s_h": (_I, [_P] * 10 + [_I, _I, _P]),
    "catre_op_gn_gelu_gemm_rows_h": (_I, [_P] * 10 + [_I, _I, _I, _P]),
    "catre_op_gnp_gelu_neck_bwd_s": (_I, [_P] * 11 + [_SZ, _I, _I, _P]),
    "catre_op_pad_cols": (_I, [_P, ctypes.c_long, ctypes.c_long, _I, _I, _P, _I, _P]),
    "catre_op_rot_l1_bwd_ws_bytes": (_SZ, [_I, _I]),
    "catre_op_rot_l1_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _I, _I, _P]),
    "catre_op_gnp_gelu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _P]),
    "catre_op_gnr_gelu_fwd": (_I, [_P, _P, _P, _P, _I, _P]),
    "catre_op_gnr_gelu_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _I, _P]),
    "catre_op_wsum_fwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "catre_op_wsum_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _P]),
    "catre_op_wsum_bwd_n": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _SZ, _I, _I, _P]),
    "catre_op_pose_update_bwd": (_I, [_P] * 13 + [_I, _P]),
    "catre_train_rot_fwd_ws_bytes": (_SZ, [_I]),
    "catre_train_rot_fwd": (_I, [_P] * 10 + [_SZ, _I, _I, _I, _I, _P]),
    "catre_op_ranger_step": (_I, [_P, _I, _P, _I, _P, _I, _P, ctypes.c_double, ctypes.c_double, ctypes.c_float,
                                  ctypes.c_float, _I, ctypes.c_float, _P]),
    "catre_aug_points": (_I, [_P] * 10 + [_I, _I, _P]),
    "catre_init_noise": (_I, [_P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _P, _P, ctypes.c_float, ctypes.c_float, _P, _I,
                              _P]),
    "catre_pcl_workspace_bytes": (_SZ, [_I, _I, _I]),
    "catre_pcl_candidates": (_I, [_P, _P, _P, _P, _P, ctypes.c_float, _I, _I, _I, _I, _P, _SZ, _P, _P]),
    "catre_pcl_sample": (_I, [_P, _P, _P, _SZ, _P, ctypes.c_uint64, _I, _I, _I, _I, _P, _P, _P]),
    "catre_pcl_fps": (_I, [_P, _P, _P, _SZ, _I, _I, _I, _I, _P, _I, _P, _P]),
    "catre_loss_fwd": (_I, [_P] * 15 + [_I, _I, _I, _P]),
    "catre_loss_bwd": (_I, [_P] * 14 + [_I, _I, _I, _P]),
    "catre_loss_fwd_sums": (_I, [_P] * 16 + [_I, _P, _I, _I, _I, _P]),
    "catre_loss_bwd_sums": (_I, [_P] * 13 + [_I, _P, _P, _P, _I, _I, _I, _P]),
    "catre_profile_enable": (_I, [_I, _I]),
    "catre_profile_collect": (_I, [_P, _I, _P]),
    "catre_debug_trunk_trace": (_I, [_P]),
    "catre_debug_knob": (_I, [_I, _I]),
    "catre_stream_capture_id": (_I, [_P, _P]),
    "catre_status_string": (ctypes.c_char_p, [_I]),
    "catre_version": (ctypes.c_char_p, []),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

KERNEL_IDS = {"stn3d": 0, "stnkd": 1, "trunk": 2, "ts_head": 3, "rot_l0_stats": 4, "rot_l1": 5, "rot_out": 6, "colmax": 7}

_lib = None

# Parameters updated through raw pointers (the fused Ranger step) do not bump torch's per-tensor version counter, so
# everything that caches a derived form of the weights (HipRuntime's packed images) also keys on this epoch.
_param_epoch = [0]


def bump_param_epoch():
    """Tell every runtime that parameters may have been rewritten out of band.

    The packed weight images are rebuilt when a parameter's ``(data_ptr, _version)`` changes - what ``load_state_dict``,
    ``p.copy_()`` and the optimizers do.  Writes through ``p.data`` (EMA updates, some third-party optimizers) bump neither;
    INFERENCE callers that do that call this once afterwards.  The training forward does not depend on it: its first
    kernel re-packs every image the forward reads on every call (``HipRuntime.train_stn3d``; with ``PCLNET.FREEZE`` the
    frozen-encoder branch of ``train_forward.forward_train`` does the same before the inference encoder kernels)."""
    _param_epoch[0] += 1


FORM_IDS = {"trunk4": 0, "stn4": 1, "stn_pair": 2, "rotw": 3, "fc_tail": 4}


def form_switch(name, value=None):
    """Kernel-form switch `name` (``FORM_IDS``) of the loaded library: set it (True / False) or just query (None);
    returns the previous setting.  All forms of a stage give the same bits - for A/B measurements and form-vs-form tests."""
    r = load().catre_form_switch(FORM_IDS[name], -1 if value is None else int(bool(value)))
    if r < 0:
        raise CatreHipError(f"unknown kernel-form switch {name!r}")
    return bool(r)


def capture_id(device):
    """Identity of the capture the current stream of ``device`` is recording into (0: not capturing)."""
    out = ctypes.c_ulonglong(0)
    check(load().catre_stream_capture_id(stream_ptr(device), ctypes.byref(out)), "catre_stream_capture_id")
    return int(out.value)


def param_epoch():
    return _param_epoch[0]


class CatreHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises ``CatreHipError`` if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CatreHipError(
            f"{LIB_PATH} not found - build it with `make -C catre_amd/csrc` (or `python -c "
            "'import __graft_entry__ as g; g.build()'`).  catre_amd has no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drifted apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().catre_status_string(status).decode()
        raise CatreHipError(f"{what} failed: {msg} (status {status})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """The current stream of `device` (a torch.device, an index or None = the current device) as a `void*`.  Through torch's
    raw-stream accessor when it exists: `torch.cuda.current_stream()` builds a Stream object per call - 5 us, ~70 times per
    training iteration."""
    if _raw_stream is not None:
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            if isinstance(device, str):
                device = torch.device(device)
            idx = device.index if device.index is not None else torch.cuda.current_device()
        return ctypes.c_void_p(_raw_stream(idx))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_dev_f32(t, name, shape=None, contiguous=True):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise CatreHipError(
            f"{name} is on {t.device}; catre_amd runs on HIP devices only (no CPU fallback - the CPU "
            "restatement lives in oracle/ and is test infrastructure)"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if shape is not None:
        if t.dim() != len(shape) or any(e is not None and int(a) != int(e) for a, e in zip(t.shape, shape)):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def points_desc(x, tfd_kps):
    """Describe the reference's ``x [B,3,N]`` / ``tfd_kps [B,3,M]`` inputs (arbitrary strides, e.g. the
    permuted views produced by ``engine/batch_test.py:91-94``) for the kernels - no copy."""
    require_dev_f32(x, "x", (None, 3, None), contiguous=False)
    require_dev_f32(tfd_kps, "tfd_kps", (x.shape[0], 3, None), contiguous=False)
    d = CatrePoints()
    d.obs, d.obs_sb, d.obs_sc, d.obs_sn = x.data_ptr(), x.stride(0), x.stride(1), x.stride(2)
    d.kps, d.kps_sb, d.kps_sc, d.kps_sn = tfd_kps.data_ptr(), tfd_kps.stride(0), tfd_kps.stride(1), tfd_kps.stride(2)
    return d


def param_array(tensors):
    """``tensors``: list in ``PARAM_KEYS`` order (``None`` allowed only for conv_p.bias)."""
    arr = (ctypes.c_void_p * CATRE_P_COUNT)()
    for i, t in enumerate(tensors):
        if t is None:
            arr[i] = None
        else:
            require_dev_f32(t, PARAM_KEYS[i])
            arr[i] = t.data_ptr()
    return arr


def profile_kernel(name, max_records):
    """Start recording HIP-event pairs around every launch of kernel ``name`` (``None`` disables)."""
    check(load().catre_profile_enable(-1 if name is None else KERNEL_IDS[name], int(max_records)), "catre_profile_enable")


def profile_collect(max_records):
    """-> list of per-launch durations in ms (synchronises on the recorded events)."""
    buf = (ctypes.c_float * max_records)()
    n = ctypes.c_int(0)
    check(load().catre_profile_collect(buf, max_records, ctypes.byref(n)), "catre_profile_collect")
    return [buf[i] for i in range(n.value)]
