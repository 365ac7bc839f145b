"""ctypes binding of libmusicgan_hip.so (the C ABI declared in include/musicgan_hip.h).

There is NO CPU or eager-PyTorch fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must be imported BEFORE the library: both link libamdhip64.so.7 and the HIP runtime that
#                             torch ships has to be the one the process binds (loading /opt/rocm's first breaks torch's device)
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmusicgan_hip.so")

MG_CONV_UPS_IN, MG_CONV_LRELU, MG_CONV_MASK_AUX, MG_CONV_PIXNORM, MG_CONV_POOL_OUT = 1, 2, 4, 8, 16
MG_CONV_MASK_OUT, MG_CONV_MASK_BYTES, MG_CONV_UNPOOL, MG_CONV_UPSUM_OUT = 32, 64, 128, 256
MG_FADE_FWD, MG_FADE_TANGENT, MG_FADE_BWD = 1, 2, 3
MG_C1_LRELU, MG_C1_TANH, MG_C1_MASK_AUX, MG_C1_TRANSPOSED, MG_C1_TANH_BWD_IN, MG_C1_ACCUM = 1, 2, 4, 8, 16, 32


class MusicGanHipError(RuntimeError):
    pass


class AdamTensor(Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("numel", c_int64), ("step_size", c_float), ("bc2_sqrt", c_float)]


class AdamTensorDev(Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p),
                ("numel", c_int64), ("step", c_void_p)]


class WgradJob(Structure):
    _fields_ = [("slab", c_void_p), ("slab_b", c_void_p), ("gw", c_void_p), ("gb", c_void_p), ("nsplit", c_int), ("Cout", c_int),
                ("Cin", c_int), ("CoutP", c_int), ("CinP", c_int), ("accumulate", c_int)]


class WgradDesc(Structure):
    _fields_ = [("x", c_void_p), ("gy", c_void_p), ("gw", c_void_p), ("gb", c_void_p), ("ws", c_void_p), ("ws_bytes", c_size_t),
                ("N", c_int), ("Cin", c_int), ("Cout", c_int), ("H", c_int), ("W", c_int), ("flags", c_int),
                ("accumulate", c_int), ("bias_n", c_int)]


class PackDesc(Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("kind", c_int), ("Co", c_int), ("Ci", c_int), ("dgrad", c_int)]


MG_PACK_CONV3X3, MG_PACK_WINO3X3, MG_PACK_UPCONV3X3, MG_PACK_UPCONV3X3_DGRAD, MG_PACK_SMALLNET, MG_PACK_WINOUPS = 0, 1, 2, 3, 4, 5

(MG_SN_LOAD, MG_SN_STORE, MG_SN_CONV, MG_SN_MASK, MG_SN_PIXNORM, MG_SN_PNBWD, MG_SN_POOL, MG_SN_POOLBWD, MG_SN_UP, MG_SN_UPBWD,
 MG_SN_LINEAR, MG_SN_LINBWD) = range(12)
MG_SN_LRELU, MG_SN_MASK_AUX, MG_SN_NOLDS = 1, 2, 4
MG_SN_MAX_OPS = 32
MG_PCM_F32, MG_PCM_I16, MG_PCM_I32, MG_PCM_U8 = 0, 1, 2, 3


class SnOp(Structure):
    _fields_ = [("op", c_int), ("src", c_int), ("dst", c_int), ("C", c_int), ("C2", c_int), ("H", c_int), ("W", c_int),
                ("flags", c_int), ("inp", c_void_p), ("aux", c_void_p), ("bias", c_void_p), ("out", c_void_p), ("out2", c_void_p)]


_P = c_void_p
# name -> (restype, argtypes); every entry must be exported by the library (checked by tests/test_abi.py)
SIGNATURES = {
    "mg_version": (c_int, []),
    "mg_last_error": (c_char_p, []),
    "mg_conv3x3_packed_floats": (c_size_t, [c_int, c_int]),
    "mg_conv3x3_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mg_conv3x3": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_wino3x3_wgrad_form": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mg_winoups3x3_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mg_winoups3x3_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "mg_winoups3x3": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_winoups3x3_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mg_winoups3x3_dgrad_pn": (c_int, [_P] * 5 + [c_int] * 5 + [c_float, _P]),
    "mg_winoups3x3_head_supported": (c_int, [c_int] * 5),
    "mg_winoups3x3_head": (c_int, [_P] * 9 + [c_int] * 5 + [c_float, _P]),
    "mg_upconv3x3_packed_floats": (c_size_t, [c_int, c_int]),
    "mg_upconv3x3_pack": (c_int, [_P, _P, c_int, c_int, _P]),
    "mg_upconv3x3": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_wino3x3_packed_floats": (c_size_t, [c_int, c_int]),
    "mg_wino3x3_pack": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mg_wino3x3": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_wino3x3_fade": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_wino3x3_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mg_wino3x3_wgrad": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "mg_wino3x3_wgrad_partial": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                         _P]),
    "mg_wino3x3_wgrad_reduce": (c_int, [_P, c_int, _P]),
    "mg_wino3x3_wgrad_partial_multi": (c_int, [_P, c_int, c_int, _P, _P]),
    "mg_conv3x3_wgrad_1x1map": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mg_conv3x3_wgrad_partial": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                         _P]),
    "mg_conv3x3_wgrad_reduce": (c_int, [_P, c_int, _P]),
    "mg_upconv3x3_dgrad_packed_floats": (c_size_t, [c_int, c_int]),
    "mg_upconv3x3_dgrad_pack": (c_int, [_P, _P, c_int, c_int, _P]),
    "mg_upconv3x3_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mg_conv3x3_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "mg_conv3x3_wgrad": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "mg_conv1x1": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_conv1x1_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mg_conv1x1_wgrad": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "mg_pixelnorm_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "mg_pixelnorm_lrelu_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "mg_upsample2x_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mg_upsample2x_bwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mg_avgpool2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mg_avgpool2_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "mg_avgpool2_bwd_tilemask": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "mg_lrelu_bwd": (c_int, [_P, _P, _P, c_size_t, c_float, _P]),
    "mg_blend_lrelu_bwd": (c_int, [_P, _P, _P, c_float, c_float, _P, _P, c_size_t, c_float, _P]),
    "mg_axpby": (c_int, [c_float, _P, c_float, _P, _P, c_size_t, _P]),
    "mg_blend_up": (c_int, [c_float, _P, c_float, _P, _P, c_int, c_int, c_int, _P]),
    "mg_axpby_dev": (c_int, [_P, _P, _P, _P, c_size_t, _P]),
    "mg_blend_up_dev": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "mg_blend_lrelu_bwd_dev": (c_int, [_P, _P, _P, _P, _P, _P, c_size_t, c_float, _P]),
    "mg_linear1_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "mg_linear1_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mg_gp_interp": (c_int, [_P, _P, _P, _P, c_int, c_size_t, _P]),
    "mg_sumsq_per_sample": (c_int, [_P, _P, c_int, c_size_t, _P]),
    "mg_scale_per_sample": (c_int, [_P, _P, _P, c_int, c_size_t, _P]),
    "mg_gp_finish": (c_int, [_P, _P, _P, c_int, c_float, c_float, _P]),
    "mg_gp_apply": (c_int, [_P, _P, _P, _P, c_int, c_size_t, c_float, c_float, _P]),
    "mg_wino3x3_mask_bytes_y_supported": (c_int, [c_int] * 5),
    "mg_stem_pair": (c_int, [_P] * 9 + [c_int] * 6 + [c_float, _P]),
    "mg_stem_pair_gx": (c_int, [_P] * 5 + [c_int] * 5 + [_P]),
    "mg_head_pair": (c_int, [_P] * 7 + [c_float, c_float] + [_P] * 3 + [c_int] * 5 + [_P]),
    "mg_blend_up_bwd": (c_int, [_P, _P, c_float, c_float, _P, _P, c_int, c_int, c_int, _P]),
    "mg_head_pair_from_mp": (c_int, [_P] * 5 + [c_float, c_float, _P, _P] + [c_int] * 4 + [_P]),
    "mg_gen_head_bwd_supported": (c_int, [c_int, c_int]),
    "mg_gen_head_bwd_supported_at": (c_int, [c_int, c_int, c_int, c_int]),
    "mg_gen_head_bwd_ws_floats": (c_size_t, [c_int, c_int, c_int]),
    "mg_gen_head_bwd": (c_int, [_P] * 10 + [c_size_t, c_int, c_int, c_int, c_float, c_int, _P]),
    "mg_channel_sum": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "mg_adam_step": (c_int, [_P, c_int, c_float, c_float, c_float, c_float, _P]),
    "mg_group_means": (c_int, [_P, c_int, c_int, _P, _P]),
    "mg_pack_multi": (c_int, [_P, c_int, _P]),
    "mg_conv3x3_small_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mg_conv3x3_small": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_conv3x3_small_pn": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_smallnet_packed_floats": (c_size_t, [c_int, c_int]),
    "mg_smallnet_buffer_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mg_smallnet": (c_int, [_P, c_int, c_int, c_int, c_size_t, c_float, _P]),
    "mg_adam_step_dev": (c_int, [_P, c_int, c_float, c_float, c_float, c_float, c_float, _P]),
    "mg_input_transform_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mg_input_transform": (c_int, [_P, c_int, _P, _P, c_size_t, c_int, c_int, c_int, c_int, c_float, _P]),
    "mg_stft_1024": (c_int, [_P, _P, _P, c_int64, _P]),
    "mg_stft_1024_pcm_ws_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "mg_stft_1024_pcm": (c_int, [_P, c_int, c_int, _P, _P, _P, c_size_t, c_int64, _P]),
    "mg_pcm_to_mono": (c_int, [_P, c_int, c_int, _P, c_int64, _P]),
    "mg_stft_generic": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "mg_crc32_f64_ws_bytes": (c_size_t, [c_int, c_int64]),
    "mg_crc32_f64": (c_int, [_P, _P, _P, c_size_t, c_int, c_int64, _P]),
    "mg_pt_write_samples": (c_int, [_P, c_int, c_int64, ctypes.c_char_p, ctypes.c_char_p, c_int64, ctypes.c_char_p, c_int64, c_int, c_int64]),
    "mg_host_io_probe": (c_int, [ctypes.c_char_p, c_int, c_int, ctypes.c_int64, ctypes.c_int64, _P, ctypes.c_int64, _P, _P]),
    "mg_codec_fwd_ws_bytes": (c_size_t, [c_int]),
    "mg_codec_fwd": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, _P]),
    "mg_codec_fwd_strided": (c_int, [_P, _P, _P, _P, c_size_t, _P, c_size_t, c_int, c_int, _P]),
    "mg_codec_inv_ws_bytes": (c_size_t, [c_int, c_int]),
    "mg_codec_inv": (c_int, [_P, _P, _P, _P, c_size_t, c_int, c_int, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises MusicGanHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MusicGanHipError(
            f"{LIB_PATH} not found: build it with `python -m musicgan_amd._build` (hipcc, gfx950). "
            "musicgan_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mg_last_error()
        raise MusicGanHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
