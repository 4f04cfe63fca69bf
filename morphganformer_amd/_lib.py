"""ctypes binding of the C ABI in include/mgf.h (libmgf_hip.so, built in-tree by morphganformer_amd/build.py).

There is NO CPU fallback: if the shared library is missing, `lib()` raises, and every op checks that its tensors
live on a HIP device.  (The reference silently falls back to slow torch ops when its plugin build fails,
torch_utils/ops/bias_act.py:39-43; a drop-in for the GPU hot path must not.)
"""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGF_LIB_PATH") or os.path.join(HERE, "libmgf_hip.so")     # env override: kernel experiments only

MGF_F32, MGF_F64, MGF_F16 = 0, 1, 2
ACT_IDS = {"linear": 1, "relu": 2, "lrelu": 3, "tanh": 4, "sigmoid": 5, "elu": 6, "selu": 7, "softplus": 8, "swish": 9,
           "relu_post": 10}      # mgf_conv1x1_f32 only: ReLU AFTER the residual add
MAX_TAPS = 9

i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class Epilogue(C.Structure):
    _fields_ = [("bias", vp), ("noise", vp), ("noise_strength", vp), ("noise_n", i32), ("act", i32), ("alpha", f32),
                ("gain", f32), ("residual", vp)]


class ConvDesc(C.Structure):
    _fields_ = [("n", i32), ("cin", i32), ("in_h", i32), ("in_w", i32), ("cout", i32), ("cout_pad", i32),
                ("tile_h", i32), ("tile_w", i32), ("istride", i32), ("ostride", i32), ("ntaps", i32), ("ngroups", i32),
                ("dy", i32 * MAX_TAPS), ("dx", i32 * MAX_TAPS), ("group", i32 * MAX_TAPS), ("oy", i32 * 4), ("ox", i32 * 4),
                ("out_h", i32), ("out_w", i32), ("y_pitch", i64), ("y_plane", i64), ("y_batch", i64), ("y_choff", i32),
                ("out_scale_stride", i32), ("workspace", vp), ("workspace_floats", i64),
                ("rgb_w", vp), ("rgb_bias", vp), ("rgb_out", vp), ("rgb_channels", i32), ("pad_", i32)]


class ConvProfRec(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("flops", f64), ("seconds", f64), ("bytes", f64), ("ksplit", i32), ("pad_", i32)]


class StyleJob(C.Structure):
    _fields_ = [("aff_w", vp), ("aff_b", vp), ("wsq", vp), ("s", vp), ("d", vp), ("cin", i32), ("cout", i32),
                ("w_offset", i32), ("aff_gain", f32), ("style_gain", f32)]


class AttnJob(C.Structure):
    _fields_ = [("wmv", vp), ("bmv", vp), ("vwb", vp), ("c", i32), ("w_offset", i32)]


class StyleBwdJob(C.Structure):
    _fields_ = [("aff_w", vp), ("wsq", vp), ("s", vp), ("d", vp), ("ds_part", vp), ("dc_part", vp), ("cin", i32), ("cout", i32),
                ("s_chunks", i32), ("d_chunks", i32), ("aff_gain", f32), ("style_gain", f32)]


class AttnGradJob(C.Structure):
    _fields_ = [("dg", vp), ("probs", vp), ("dc", vp), ("cpre", vp), ("part", vp), ("dvwb", vp), ("dc_part", vp), ("c", i32), ("f", i32),
                ("slices", i32), ("nchunk", i32), ("blk_grad", i32), ("blk_dot", i32), ("blk_red", i32), ("pad_", i32)]


class AttnBwdJob(C.Structure):
    _fields_ = [("wmv", vp), ("dvwb", vp), ("c", i32), ("pad_", i32)]


_SIGS = {
    "mgf_last_error": (C.c_char_p, []),
    "mgf_version": (C.c_int, []),
    "mgf_device_ok": (C.c_int, []),
    "mgf_bias_act": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, i64, i64, i64, C.c_int, C.c_int, f32, f32, f32, vp]),
    "mgf_upfirdn2d": (C.c_int, [vp, vp, vp, C.c_int, i32, i32, i32, i32, i64, i64, i64, i64, i32, i32, i64, i64, i64, i64,
                                i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, C.POINTER(Epilogue), vp]),
    "mgf_conv_taps_f32": (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(ConvDesc), C.POINTER(Epilogue), vp]),
    "mgf_conv_taps_bf16x3_f32": (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(ConvDesc), C.POINTER(Epilogue), vp]),
    "mgf_tconv3x3s2_border_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i64, i64, i64, vp]),
    "mgf_winograd_weights_f32": (C.c_int, [vp, vp, i32, i32, f32, vp]),
    "mgf_conv3x3_winograd_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Epilogue), vp]),
    "mgf_winograd2_weights_f32": (C.c_int, [vp, vp, i32, i32, f32, vp]),
    "mgf_conv3x3_winograd2_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Epilogue), vp]),
    "mgf_conv3x3_winograd2_rgb_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mgf_conv3x3_winograd3_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Epilogue), vp]),
    "mgf_conv1x1_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i32, C.POINTER(Epilogue), vp]),
    "mgf_conv3x3s2_few_inputs_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mgf_tconv3x3s2_few_outputs_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, vp]),
    "mgf_conv1x1_force_shape": (C.c_int, [i32]),
    "mgf_conv3x3_winograd3_slice_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, C.POINTER(Epilogue), vp]),
    "mgf_conv3x3_winograd3_up2res_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.POINTER(Epilogue), vp]),
    "mgf_winograd3_force_shape": (C.c_int, [i32]),
    "mgf_conv3x3_winograd3_rgb_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "mgf_conv3x3_winograd2_slice_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, i32, C.POINTER(Epilogue), vp]),
    "mgf_pack_conv_weights": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]),
    "mgf_conv_profile_begin": (C.c_int, []),
    "mgf_conv_profile_end": (C.c_int, [C.POINTER(ConvProfRec), i32]),
    "mgf_demod_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "mgf_demod_bwd_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mgf_style_demod": (C.c_int, [C.POINTER(StyleJob), vp, i64, i32, i32, vp]),
    "mgf_style_demod_multi": (C.c_int, [vp, i32, vp, i64, i32, i32, i32, vp]),
    "mgf_duplex_attention": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, C.POINTER(Epilogue), i32, vp, vp, vp]),
    "mgf_att_map_upsample_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mgf_attn_values": (C.c_int, [C.POINTER(AttnJob), vp, i64, i64, i32, i32, i32, vp]),
    "mgf_attn_values_multi": (C.c_int, [vp, i32, vp, i64, i64, i32, i32, i32, vp]),
    "mgf_randn_f32": (C.c_int, [vp, i64, C.c_uint64, vp, vp]),
    "mgf_rgb_weights_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "mgf_mapping_param_floats": (i64, [i32, i32, i32]),
    "mgf_mapping_forward": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_reduce_scratch_floats": (i64, []),
    "mgf_mse_f32": (C.c_int, [vp, vp, vp, i32, i64, i64, f32, i32, vp, vp]),
    "mgf_dssim_scratch_bytes": (i64, [i32, i32, i32, i32]),
    "mgf_dssim_u8_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i64, f32, f32, i32, vp, vp]),
    "mgf_lbp_scratch_bytes": (i64, [i32]),
    "mgf_lbp_gray224_u8": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mgf_lbp_codes_u8": (C.c_int, [vp, vp, vp, i32, vp]),
    "mgf_lbp_distance_f64": (C.c_int, [vp, vp, vp, vp, i32, vp, vp]),
    "mgf_wing_loss_f64": (C.c_int, [vp, vp, vp, i32, i64, f64, f64, vp, i32, vp]),
    "mgf_adaptive_wing_loss_f64": (C.c_int, [vp, vp, vp, i32, i64, f64, f64, f64, f64, vp, i32, vp]),
    "mgf_lpips_unit_f32": (C.c_int, [vp, vp, i32, i32, i64, vp]),
    "mgf_lpips_layer_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i64, i64, i32, vp, vp]),
    "mgf_lpips_stem_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "mgf_channel_affine_prelu_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, vp]),
    "mgf_linear_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "mgf_resize_bilinear_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_spatial_mean_f32": (C.c_int, [vp, vp, i32, i64, vp]),
    "mgf_l2_normalize_f32": (C.c_int, [vp, vp, i32, i32, f32, vp]),
    "mgf_l2_normalize_bwd_f32": (C.c_int, [vp, vp, vp, i32, i32, f32, vp]),
    "mgf_spatial_mean_bwd_f32": (C.c_int, [vp, vp, i32, i64, vp]),
    "mgf_relu_bwd_slice_f32": (C.c_int, [vp, vp, i32, i32, vp, i32, i32, i32, i32, i64, vp]),
    "mgf_maxpool_s2_floor_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "mgf_maxpool3x3s2_ceil_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_latent_perturb": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, vp]),
    "mgf_latent_perturb_mean": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    "mgf_select_best": (C.c_int, [vp, vp, vp, vp, vp, i64, vp, vp, vp, f64, f32, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp]),
    "mgf_keep_improvements": (C.c_int, [vp, vp, i64, vp, i32, vp]),
    "mgf_to_uint8_hwc": (C.c_int, [vp, vp, i32, i32, i32, vp]),
    "mgf_reference_gray_scratch_floats": (i64, []),
    "mgf_reference_gray_u8": (C.c_int, [vp, vp, i32, i32, i32, vp, vp]),
    "mgf_bwd_chunks": (i32, [i64]),
    "mgf_layer_act_bwd_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, f32, f32, vp]),
    "mgf_channel_dot_f32": (C.c_int, [vp, vp, vp, i32, i32, i64, vp]),
    "mgf_style_grad_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    "mgf_style_grad_act_bwd_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i64, f32, f32, vp]),
    "mgf_style_act_fir_tiles": (i32, [i32, i32]),
    "mgf_style_act_fir_bwd_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, f32, i32, i32, i32, i32, f32, f32, vp]),
    "mgf_layer_act_bwd_low_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i64, f32, f32, vp]),
    "mgf_duplex_attention_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mgf_attn_values_grad": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mgf_attn_values_grad_workspace_floats": (C.c_int64, [i32, i32]),
    "mgf_attn_values_grad_ws": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp, i64, vp]),
    "mgf_style_demod_bwd_multi": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "mgf_attn_values_bwd_multi": (C.c_int, [vp, vp, i32, i32, i32, i32, vp]),
    "mgf_attn_values_grad_slices": (i32, [i32, i32, i32, i32]),
    "mgf_attn_grad_job_bytes": (i64, []),
    "mgf_attn_grad_multi": (C.c_int, [vp, i32, i32, i32, i32, i32, vp]),
    "mgf_latent_bwd_multi": (C.c_int, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_latent_grad_gather": (C.c_int, [vp, vp, i32, vp, i32, i32, i32, i32, f32, vp]),
    "mgf_lpips_layer_bwd_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i64, i64, f32, i32, vp]),
    "mgf_lpips_layer_bwd_relu_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, f32, vp]),
    "mgf_lpips_layer_stats_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, i64, i32, vp, vp]),
    "mgf_lpips_layer_defer_f32": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i64, i64, C.POINTER(i32), vp]),
    "mgf_lpips_finish_taps_f32": (C.c_int, [vp, vp, i64, i32, C.POINTER(i32), C.POINTER(f32), i32, i32, vp]),
    "mgf_lpips_layer_bwd_relu_stats_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, i64, f32, vp]),
    "mgf_relu_bwd_split_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i64, vp]),
    "mgf_maxpool3x3s2_ceil_idx_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_maxpool3x3s2_ceil_bwd_idx_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_maxpool3x3s2_ceil_bwd_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_maxpool_s2_floor_bwd_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "mgf_mse_grad_f32": (C.c_int, [vp, vp, vp, i32, i64, i64, f32, i32, vp]),
    "mgf_prelu_bwd_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i64, vp]),
    "mgf_linear_bwd_f32": (C.c_int, [vp, vp, vp, i32, i32, i32, vp]),
    "mgf_resize_bilinear_bwd_f32": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_cv_warp_triangle_bytes": (i64, []),
    "mgf_cv_warp_triangles_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp]),
    "mgf_adam_step_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, vp]),
    "mgf_mapping_bwd_scratch_floats": (i64, [i32, i32, i32]),
    "mgf_mapping_backward": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_mapping_forward_save": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "mgf_mapping_backward_saved": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
}

EXPORTED_SYMBOLS = sorted(_SIGS)
_lib = None


class MgfError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libmgf_hip.so (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgfError(f"{LIB_PATH} is missing: run `python -m morphganformer_amd.build` (or __graft_entry__.build()). "
                           "There is no CPU fallback for the HIP path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mgf_last_error().decode(errors="replace")
        raise MgfError(f"{what or 'mgf'} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and t.device.type != "cuda":
            raise MgfError(f"HIP op received a tensor on '{t.device}'; the MI355X path has no CPU fallback "
                           "(use impl='ref' only in oracle tests)")


def dtype_id(dt: torch.dtype) -> int:
    try:
        return {torch.float32: MGF_F32, torch.float64: MGF_F64, torch.float16: MGF_F16}[dt]
    except KeyError:
        raise MgfError(f"unsupported dtype {dt}") from None


def make_epilogue(bias=None, noise=None, noise_strength=None, noise_n=1, act="linear", alpha=0.2, gain=1.0, residual=None):
    ep = Epilogue(ptr(bias), ptr(noise), ptr(noise_strength), int(noise_n), ACT_IDS[act], float(alpha), float(gain),
                  ptr(residual))
    # the struct holds raw device pointers: keep the tensors alive as long as it lives, so that a temporary passed by the caller
    # (`make_epilogue(bias=b.cuda())`) is not handed back to the caching allocator -- and reused for the launch's own output -- before
    # the kernel that reads it has been enqueued (after that the allocator's stream ordering protects it)
    ep._keep = (bias, noise, noise_strength, residual)
    return ep
