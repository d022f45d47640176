"""ctypes binding of libcfpnet_hip.so (the C ABI declared in include/cfpnet_hip.h).

The product path has no CPU fallback: if the library is missing or a symbol is absent this
module raises at import/first use, and every non-zero return code becomes a RuntimeError
carrying `cfp_last_error()`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFP_HIP_LIB") or os.path.join(_HERE, "libcfpnet_hip.so")      # CFP_HIP_LIB: A/B runs against another build

F32, BF16, F16 = 0, 1, 2
F32X3 = 3      # float32 storage + f16x3 matrix math: a planning / packing dtype (include/cfpnet_hip.h)
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SILU, ACT_GELU, ACT_SIGMOID = range(6)
TOF_SAMPLE_UNIFORM, TOF_SAMPLE_ICDF = 0, 1
HEAD_WOUT_HILO, HEAD_RAM_HILO = 1, 2
CONV_PER_IMAGE, CONV_W2, CONV_IN_FLIGHT, CONV_X3, CONV_WS_TICKETS = 1, 2, 4, 8, 16
CONV_TICKET_BYTES = 4096     # CFP_CONV_TICKET_BYTES

_p, _i, _f, _sz, _ll = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong

# name -> (restype, argtypes); must list every symbol of include/cfpnet_hip.h
SIGNATURES = {
    "cfp_version": (_i, []),
    "cfp_last_error": (C.c_char_p, []),
    "cfp_conv2d_nhwc": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i] + [_i] * 14 + [_p, _sz, _p]),
    "cfp_conv2d_nhwc_ex": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i] + [_i] * 14 + [_p, _p, _f, _i, _p, _sz, _p]),
    "cfp_conv2d_nhwc_moments": (_i, [_p, _i, _p, _p, _p, _i] + [_i] * 13 + [_p, _sz, _p, _sz, _p, _p, _p]),
    "cfp_conv2d_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "cfp_pack_w_x3_elems": (_sz, [_ll, _i]),
    "cfp_pack_w_x3": (_i, [_p, _p, _ll, _i, _p]),
    "cfp_conv2d_plan": (_i, [_i] * 8 + [_p, _p]),
    "cfp_debug_set": (_i, [_i, _i]),
    "cfp_conv2d_variant": (_i, [_i, _i]),
    "cfp_dwconv3x3_nhwc": (_i, [_p, _i, _p, _p, _p, _p, _i] + [_i] * 11 + [_p]),
    "cfp_dwconv3x3_strips": (_i, [_i] * 6),
    "cfp_dwconv3x3_launch_slots": (_i, [_i] * 10),
    "cfp_dwconv3x3_sum_nhwc": (_i, [_p, _i, _p, _p, _p, _p, _i, _p] + [_i] * 11 + [_p]),
    "cfp_se_fold": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_se_gate_fold": (_i, [_p, _i, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_dwconv3x3_se_parts": (_i, [_i, _i, _i, _i, _i, _i]),
    "cfp_dwconv3x3_se_nhwc": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _p] + [_i] * 11 + [_p]),
    "cfp_se_gate_fold2": (_i, [_p, _i, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_dwconv_large_nhwc": (_i, [_p, _i, _p, _p, _p, _p, _i] + [_i] * 7 + [_p]),
    "cfp_dwconv_large_toeplitz_elems": (_sz, [_i, _i]),
    "cfp_dwconv_large_toeplitz": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "cfp_dwconv_large_mfma_nhwc": (_i, [_p, _i, _p, _p, _p, _p, _i] + [_i] * 7 + [_p]),
    "cfp_channel_sum": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_se_hidden": (_i, [_p, _i, _f, _p, _p, _p, _i, _i, _i, _p]),
    "cfp_se_scale": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_scale_channels": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "cfp_layernorm": (_i, [_p, _i, _p, _p, _f, _p, _i, _p, _i, _i, _i, _i, _p]),
    "cfp_attn_kv_ws_floats": (_sz, [_i] * 7),
    "cfp_attn_kv_reduce": (_i, [_p, _i, _p, _i, _p, _p, _p] + [_i] * 10 + [_f, _i, _i, _i, _p]),
    "cfp_attn_apply": (_i, [_p, _i, _p, _p, _p, _i] + [_i] * 9 + [_f, _f, _i, _i, _i, _p]),
    "cfp_loftr_tail": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f] + [_i] * 5 + [_f, _f, _i, _i, _i, _p]),
    "cfp_resize_bilinear": (_i, [_p, _i] + [_i] * 6 + [_p, _i] + [_i] * 6 + [_p] + [_i] * 7 + [_p]),
    "cfp_add_rowtable": (_i, [_p, _i, _p, _p, _i] + [_i] * 8 + [_p]),
    "cfp_upsample_cat_conv3x3": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _i] + [_i] * 6 + [_p]),
    "cfp_copy_rows": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "cfp_copy_rows2": (_i, [_p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _p]),
    "cfp_rgb_to_nhwc8": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "cfp_rgb_to_nhwc8_hilo": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "cfp_scalar_to_rows8": (_i, [_p, _p, _i, _i, _p]),
    "cfp_silog_ws_bytes": (_sz, [_i, _i, _i]),
    "cfp_silog_loss_fwd": (_i, [_p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "cfp_silog_loss_bwd": (_i, [_p, _p, _f, _i, _i, _i, _i, _i, _i, _p, _p]),
    "cfp_adamw_step": (_i, [_p, _p, _p, _p, C.c_longlong, _f, _f, _f, _f, _f, _i, _p, _p]),
    "cfp_grad_clip_ws_bytes": (_sz, []),
    "cfp_grad_clip_factor": (_i, [_p, C.c_longlong, _f, _p, _sz, _p, _p]),
    "cfp_tof_hist_sim": (_i, [_p, C.c_longlong, _i, _i, _i, _i, _i, _i, _i, _p, _i, _f, _i, C.c_double, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p]),
    "cfp_tof_sample_points": (_i, [_p, _p, _p, _p, C.c_longlong, _i, _i, _p, _p]),
    "cfp_eval_metrics_ws_bytes": (_sz, [_i]),
    "cfp_eval_metrics": (_i, [_p, _i, _i, _p, _i, _i, _i, _i, _i, _f, _f, _p, _sz, _p, _p]),
    "cfp_conv2d_wgrad_ws_bytes": (_sz, [_i, _i, _i]),
    "cfp_conv2d_wgrad": (_i, [_p, _i, _p, _i, _p] + [_i] * 12 + [_f, _i, _p, _sz, _p]),
    "cfp_conv2d_wgrad_bias": (_i, [_p, _i, _p, _i, _p, _p] + [_i] * 12 + [_f, _f, _i, _p, _sz, _p]),
    "cfp_conv2d_wgrad_deferred": (_i, [_p, _i, _p, _i, _p, _p] + [_i] * 12 + [_f, _f, _i, _p, _sz, _p, _p]),
    "cfp_wgrad_reduce_jobs": (_i, [_p, _i, _p]),
    "cfp_conv2d_weight_flip": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_weight_flip_blocks": (_i, [_ll]),
    "cfp_conv2d_weight_flip_batch": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "cfp_conv2d_dgrad": (_i, [_p, _i, _p, _p, _i] + [_i] * 14 + [_p, _sz, _p]),
    "cfp_bn_ws_bytes": (_sz, [_i]),
    "cfp_bn_train_stats": (_i, [_p, _i, C.c_longlong, _i, _i, _p, _p, _f, _f] + [_p] * 7 + [_p, _sz, _p]),
    "cfp_bn_train_stats_partials": (_i, [_p, _i, C.c_longlong, C.c_longlong, _i, _p, _p, _f, _f] + [_p] * 7 + [_p]),
    "cfp_scale_shift_act": (_i, [_p, _i, _p, _p, _i, _p, _i, C.c_longlong, _i, _i, _p]),
    "cfp_scale_shift_act_res": (_i, [_p, _i, _p, _p, _i, _p, _i, _p, _i, C.c_longlong, _i, _i, _p]),
    "cfp_bn_train_bwd": (_i, [_p, _i, _p, _i, C.c_longlong, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _i, _p, _sz, _p]),
    "cfp_colsum": (_i, [_p, _i, C.c_longlong, _i, _i, _p, _p, _sz, _p]),
    "cfp_act_bwd": (_i, [_p, _i, _p, _i, _i, _p, _i, C.c_longlong, _i, _i, _p]),
    "cfp_layernorm_bwd_ws_bytes": (_sz, [C.c_longlong, _i]),
    "cfp_layernorm_bwd": (_i, [_p, _i, _p, _i, _p, _f, _p, _i, _i, _p, _p, C.c_longlong, _i, _i, _p, _sz, _p]),
    "cfp_layernorm_bwd_deferred": (_i, [_p, _i, _p, _i, _p, _f, _p, _i, _i, _p, _p, C.c_longlong, _i, _i, _p, _sz, _p, _p]),
    "cfp_axpby": (_i, [_p, _i, _p, _i, _f, _f, _p, _i, C.c_longlong, _i, _i, _p]),
    "cfp_rowtable_grad": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "cfp_se_train_ws_floats": (_sz, [_i, _i, _i]),
    "cfp_se_train_fwd": (_i, [_p, _i, _f] + [_p] * 7 + [_i, _i, _i, _p]),
    "cfp_se_train_bwd": (_i, [_p] * 12 + [_f, _f, _i, _i, _i, _p]),
    "cfp_channel_dot": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _p]),
    "cfp_bcast_fma": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_dwconv3x3_dgrad": (_i, [_p, _i, _p, _p, _i] + [_i] * 11 + [_p]),
    "cfp_dwconv3x3_wgrad_ws_bytes": (_sz, [_i]),
    "cfp_dwconv3x3_wgrad": (_i, [_p, _i, _p, _i, _p] + [_i] * 9 + [_f, _i, _p, _sz, _p]),
    "cfp_dwconv3x3_wgrad_deferred": (_i, [_p, _i, _p, _i, _p] + [_i] * 9 + [_f, _i, _p, _sz, _p, _p]),
    "cfp_index_rows": (_i, [_p, _i, _p, _p, _i, C.c_longlong, _i, _i, _i, _p]),
    "cfp_resize_bilinear_bwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "cfp_bin_centers": (_i, [_p, _f, _f, _p, _p, _i, _i, _p]),
    "cfp_bin_centers_bwd": (_i, [_p, _f, _f, _p, _i, _i, _p]),
    "cfp_softmax_expect_ws_bytes": (_sz, [_i, _i, _i]),
    "cfp_softmax_expect": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "cfp_linattn_state_bytes": (_sz, [_i, _i, _i]),
    "cfp_linattn_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "cfp_linattn_fwd": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _f, _i, _p, _sz, _p]),
    "cfp_linattn_bwd": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p, _sz, _p]),
    "cfp_dwconv_large_wgrad_ws_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "cfp_dwconv_large_wgrad": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _f, _i, _p, _sz, _p]),
    "cfp_row_normalize": (_i, [_p, _p, _p, _i, _i, _p]),
    "cfp_add_rowtable_dev": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "cfp_rowtable_grad_dev": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _f, _i, _p]),
    "cfp_nyu_augment": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p, _p, _p]),
    "cfp_nyu_rotate": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "cfp_bin_regressor": (_i, [_p, _i, _f] + [_p] * 7 + [_f, _f, _i, _p, _p, _i, _i, _i, _i, _p]),
    "cfp_bin_softmax": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "cfp_conv3x3_mean": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "cfp_lkpm_tail": (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _f, _i, _i, _i, _p]),
    "cfp_mbconv_plan": (_i, [_i, _i, _i, _i, _i, _p, _p]),
    "cfp_mbconv_expand_dw": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "cfp_hist_encoder": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "cfp_depth_head_fused": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "cfp_bin_head_fused": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
}

class WgradJob(C.Structure):
    """cfp_wgrad_job of include/cfpnet_hip.h."""
    _fields_ = [("slabs", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("n", C.c_longlong), ("n_dw", C.c_longlong),
                ("nsplit", C.c_int), ("ew", C.c_int), ("beta", C.c_float), ("beta_b", C.c_float)]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            f"(`python -c 'import __graft_entry__ as g; g.build()'` or `make -C cfpnet_amd/csrc`). "
            "There is no CPU fallback for the product path.")
    # torch FIRST: its wheel carries its own libamdhip64; the kernels must run in the HIP runtime that owns the tensors they are handed.  Loaded the
    # other way round (library, then torch -- `build()` followed by `smoke()` in one process) the process ends up with two runtimes and the first
    # kernel launch fails with "no ROCm-capable device is detected" (seen on the GPU box in round 5).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().cfp_last_error().decode()


def call(name: str, *args):
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {last_error()}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
