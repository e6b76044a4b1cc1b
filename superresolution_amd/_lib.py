"""ctypes binding of librestormer_hip.so (the C ABI in include/restormer_hip.h).

The library is the only compute path of this package: if it is missing or a
call fails, a RuntimeError is raised -- there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RST_LIB_PATH: another build of the same library, for A/B timing of kernel variants)
LIB_PATH = os.environ.get("RST_LIB_PATH") or os.path.join(_HERE, "librestormer_hip.so")

ABI_VERSION = 1


class BlockDesc(ctypes.Structure):
    """rst_block_desc"""
    _fields_ = [(n, c_int32) for n in ("B", "C", "H", "W", "heads", "hidden", "ln_with_bias", "conv_bias")]


PARAM_FIELDS = (
    "norm1_weight", "norm1_bias", "temperature", "qkv_weight", "qkv_bias", "qkv_dw_weight", "qkv_dw_bias",
    "proj_weight", "proj_bias", "norm2_weight", "norm2_bias", "pin_weight", "pin_bias", "ffn_dw_weight",
    "ffn_dw_bias", "pout_weight", "pout_bias",
)


class BlockParams(ctypes.Structure):
    """rst_block_params"""
    _fields_ = [(n, c_void_p) for n in PARAM_FIELDS]


class BlockGrads(ctypes.Structure):
    """rst_block_grads"""
    _fields_ = [(n, c_void_p) for n in PARAM_FIELDS]


_P = c_void_p
_DESC = POINTER(BlockDesc)
_PAR = POINTER(BlockParams)
_GR = POINTER(BlockGrads)

# name -> (restype, argtypes); every symbol declared in include/restormer_hip.h
SIGNATURES = {
    "rst_version": (c_int32, []),
    "rst_bwd_side_stream_enabled": (c_int32, []),
    "rst_set_exact_fp32": (None, [c_int32]),
    "rst_get_exact_fp32": (c_int32, []),
    "rst_last_error": (c_char_p, []),
    "rst_layernorm_fwd": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int64, _P]),
    "rst_layernorm_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int64]),
    "rst_layernorm_bwd": (c_int32, [_P, _P, c_int32, _P, _P, c_int32, _P, _P, c_int32, c_int32, c_int64, _P, c_size_t, _P]),
    "rst_layernorm_bwd_noalloc": (c_int32, [_P, _P, c_int32, _P, _P, c_int32, _P, _P, c_int32, c_int32, c_int64, _P]),
    "rst_conv1x1_fwd": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, _P]),
    "rst_conv1x1_cat_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rst_conv1x1_cat_fwd": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int64, _P, c_size_t, _P]),
    "rst_conv1x1_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int64]),
    "rst_conv1x1_bwd": (c_int32, [_P, _P, _P, _P, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int64, _P, c_size_t, _P]),
    "rst_dwconv3x3_fwd": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_conv3x3_fewout_fwd": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_conv3x3_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "rst_conv3x3_fwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "rst_conv3x3_supported": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "rst_conv3x3_fwd": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "rst_conv3x3_fewout_up2_fwd": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_dwconv3x3_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "rst_dwconv3x3_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "rst_pixel_unshuffle2": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_pixel_shuffle2": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_attention_workspace_bytes": (c_size_t, [_DESC]),
    "rst_attention_saved_bytes": (c_size_t, [_DESC]),
    "rst_attention_fwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _P, c_size_t, _P]),
    "rst_attention_bwd_workspace_bytes": (c_size_t, [_DESC]),
    "rst_attention_bwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, c_int32, _GR, _P, c_size_t, _P]),
    "rst_ffn_workspace_bytes": (c_size_t, [_DESC]),
    "rst_ffn_saved_bytes": (c_size_t, [_DESC]),
    "rst_ffn_fwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _P, c_size_t, _P]),
    "rst_ffn_bwd_workspace_bytes": (c_size_t, [_DESC]),
    "rst_ffn_bwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, c_int32, _GR, _P, c_size_t, _P]),
    "rst_block_workspace_bytes": (c_size_t, [_DESC]),
    "rst_block_saved_bytes": (c_size_t, [_DESC]),
    "rst_block_fwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, c_size_t, _P]),
    "rst_block_prepared_bytes": (c_size_t, [_DESC]),
    "rst_block_prepare": (c_int32, [_DESC, _PAR, _P, _P]),
    "rst_block_fwd_prepared": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, c_size_t, _P]),
    "rst_block_bwd_workspace_bytes": (c_size_t, [_DESC]),
    "rst_block_shard_workspace_bytes": (c_size_t, [_DESC]),
    "rst_block_shard_stats_floats": (c_size_t, [_DESC]),
    "rst_block_shard_fwd_stats": (c_int32, [_DESC, _PAR, _P, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "rst_block_shard_fwd_apply": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, c_size_t, _P]),
    "rst_bf16_packed_bytes": (c_size_t, [c_int32, c_int32, c_int64]),
    "rst_bf16_pack": (c_int32, [_P, c_int32, _P, c_int32, c_int32, c_int64, _P]),
    "rst_bf16_unpack": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int64, _P]),
    "rst_bf16_to_channels_last": (c_int32, [_P, _P, c_int32, c_int32, c_int64, _P]),
    "rst_bf16_from_channels_last": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_bf16_conv1x1_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "rst_bf16_conv1x1_fwd": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, _P, c_size_t, _P]),
    "rst_bf16_conv3x3_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "rst_bf16_conv3x3_supported": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "rst_bf16_conv3x3_fwd": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "rst_bf16_conv3x3_fewout_fwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_bf16_block_workspace_bytes": (c_size_t, [_DESC]),
    "rst_bf16_block_prepared_bytes": (c_size_t, [_DESC]),
    "rst_bf16_block_prepare": (c_int32, [_DESC, _PAR, _P, _P]),
    "rst_bf16_block_fwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, c_size_t, _P]),
    "rst_bf16_block_bwd_workspace_bytes": (c_size_t, [_DESC]),
    "rst_bf16_block_bwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _GR, _P, c_size_t, _P]),
    "rst_bf16_block_kept_bytes": (c_size_t, [_DESC]),
    "rst_bf16_block_fwd_keep": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _P, c_size_t, _P]),
    "rst_bf16_block_bwd_kept": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _GR, _P, _P, c_size_t, _P]),
    "rst_bf16_block_saved_bytes": (c_size_t, [_DESC]),
    "rst_bf16_block_bwd_saved_workspace_bytes": (c_size_t, [_DESC]),
    "rst_bf16_block_fwd_save": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _P, c_size_t, _P]),
    "rst_bf16_block_bwd_saved": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _GR, _P, _P, c_size_t, _P]),
    "rst_bf16_conv3x3_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "rst_bf16_conv3x3_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "rst_bf16_conv1x1_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int64]),
    "rst_bf16_conv1x1_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int64, _P, c_size_t, _P]),
    "rst_bf16_pixel_shuffle2": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_bf16_pixel_unshuffle2": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_bf16_copy_channel_blocks": (c_int32, [_P, c_int32, c_int32, _P, c_int32, c_int32, c_int32, c_int32, c_int64, _P]),
    "rst_conv3x3_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "rst_conv3x3_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "rst_raw_prepare": (c_int32, [_P, c_int32, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "rst_profile_begin": (c_int32, []),
    "rst_profile_end": (c_int32, [c_char_p, c_size_t]),
    "rst_peak_copy": (c_int32, [_P, _P, c_size_t, c_int32, _P]),
    "rst_peak_mfma": (c_int32, [c_int32, c_int32, c_int32, _P, _P]),
    "rst_block_bwd": (c_int32, [_DESC, _PAR, _P, _P, _P, _P, _GR, _P, c_size_t, _P]),
}

_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library (once) and attach the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C superresolution_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    # torch must be imported first so that this library binds to the HIP runtime torch already loaded
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.rst_version() != ABI_VERSION:
        raise RuntimeError(f"librestormer_hip.so ABI {lib.rst_version()} != expected {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().rst_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rst_status {status}): {msg}")
