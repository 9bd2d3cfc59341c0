"""ctypes binding of libmi_depth.so (include/mi_depth.h).

The HIP library is the product: if it is missing the import fails loudly -- there is no
CPU/PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi_depth.so")

MD_OK = 0
MD_ERR_INVALID_ARG, MD_ERR_SHAPE, MD_ERR_IO, MD_ERR_FORMAT, MD_ERR_HIP = -1, -2, -3, -4, -5
MD_ERR_UNSUPPORTED, MD_ERR_NO_FOV, MD_ERR_OOM, MD_ERR_LEVELS = -6, -7, -8, -9
MD_MEM_HOST, MD_MEM_DEVICE = 0, 1
MD_COMM_ID_BYTES = 128
TILE_256x256, TILE_128x128, TILE_256x32, TILE_128x64, TILE_64x64, TILE_AUTO = 0, 1, 2, 3, 4, 99


class MdError(RuntimeError):
    """Maps the reference's `Result::Err(String)` / `RecorderError` / panics."""

    def __init__(self, code: int, message: str):
        super().__init__(f"mi_depth error {code}: {message}")
        self.code = code
        self.message = message


class MdDa3Outputs(C.Structure):
    """md_da3_outputs (include/mi_depth.h)."""
    _fields_ = [("depth", C.c_void_p), ("depth_confidence", C.c_void_p), ("aux", C.c_void_p), ("aux_confidence", C.c_void_p),
                ("pose_encoding", C.c_void_p), ("extrinsics", C.c_void_p), ("intrinsics", C.c_void_p)]


class MdDa3Cfg(C.Structure):
    _fields_ = [("variant", C.c_char_p), ("image_size", C.c_int), ("precision", C.c_int), ("max_batch", C.c_int),
                ("ln_eps", C.c_float), ("image_width", C.c_int)]


class MdNchwView(C.Structure):
    """md_nchw_view (include/mi_depth.h): one NCHW fp32 tensor with its shape."""
    _fields_ = [("data", C.c_void_p), ("channels", C.c_int), ("height", C.c_int), ("width", C.c_int)]


class MdHeadDebug(C.Structure):
    """md_head_debug (include/mi_depth.h) = `HeadDebug`, depth_pro/mod.rs:135-142."""
    _fields_ = [("conv0", C.c_void_p), ("deconv", C.c_void_p), ("conv1", C.c_void_p), ("relu", C.c_void_p),
                ("pre_out", C.c_void_p), ("canonical", C.c_void_p)]


class MdDepthProCfg(C.Structure):
    _fields_ = [
        ("patch_encoder_preset", C.c_char_p),
        ("image_encoder_preset", C.c_char_p),
        ("fov_encoder_preset", C.c_char_p),
        ("decoder_features", C.c_int),
        ("use_fov_head", C.c_int),
        ("interpolation", C.c_int),
        ("precision", C.c_int),
        ("max_batch", C.c_int),
        ("ln_eps", C.c_float),
    ]


# every symbol include/mi_depth.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
_I = C.c_int
SYMBOLS = {
    "md_last_error": (C.c_char_p, []),
    "md_version": (C.c_char_p, []),
    "md_device_open": (_I, [_I, C.POINTER(_P)]),
    "md_device_close": (_I, [_P]),
    "md_device_synchronize": (_I, [_P]),
    "md_depth_pro_cfg_default": (None, [C.POINTER(MdDepthProCfg)]),
    "md_depth_pro_create": (_I, [_P, C.POINTER(MdDepthProCfg), C.c_uint64, _I, C.POINTER(_P)]),
    "md_depth_pro_load": (_I, [_P, C.c_char_p, C.POINTER(_P)]),
    "md_depth_pro_load_with_config": (_I, [_P, C.POINTER(MdDepthProCfg), C.c_char_p, C.POINTER(_P)]),
    "md_checkpoint_info": (_I, [C.c_char_p, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(_I), C.POINTER(C.c_int64 * 8), C.POINTER(_I)]),
    "md_checkpoint_read_tensor": (_I, [C.c_char_p, C.c_char_p, _P, C.c_size_t]),
    "md_model_set_tensor": (_I, [_P, C.c_char_p, _P, C.c_size_t]),
    "md_model_get_tensor": (_I, [_P, C.c_char_p, _P, C.c_size_t]),
    "md_model_param_count": (_I, [_P]),
    "md_model_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]),
    "md_model_commit_weights": (_I, [_P]),
    "md_model_round_weights_f16": (_I, [_P]),
    "md_model_weight_arena": (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "md_model_destroy": (_I, [_P]),
    "md_model_fork": (_I, [_P, C.POINTER(_P)]),
    "md_depth_pro_infer": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "md_depth_pro_decoder_from_features": (_I, [_P, C.POINTER(MdNchwView), _I, _I, _I, _P, _P, C.POINTER(C.c_void_p), _I, _P]),
    "md_depth_pro_head_debug": (_I, [_P, C.POINTER(MdNchwView), _I, _I, C.POINTER(MdHeadDebug), _I, _P]),
    "md_depth_pro_infer_windows": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    "md_infer_from_rgb": (_I, [_P, _P, C.c_size_t, _I, _I, _I, _P, _P, _P, _I, _P]),
    "md_da3_cfg_default": (None, [C.POINTER(MdDa3Cfg)]),
    "md_da3_create": (_I, [_P, C.POINTER(MdDa3Cfg), C.c_uint64, _I, C.POINTER(_P)]),
    "md_da3_load": (_I, [_P, C.POINTER(MdDa3Cfg), C.c_char_p, C.POINTER(_P)]),
    "md_da3_infer": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "md_da3_infer_ex": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "md_model_enable_graph": (_I, [_P, _I]),
    "md_da3_infer_with_camera": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P]),
    "md_gemm_ksplit_launches": (_I, []),
    "md_debug_gemm_direct_store": (_I, [_I]),
    "md_debug_gemm_persistent": (_I, [_I]),
    "md_debug_gemm_stagger": (_I, [_I, _I]),
    "md_da3_infer_raw": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _P]),
    "md_da3_infer_from_tokens": (_I, [_P, C.POINTER(C.c_void_p), _I, _I, _I, _I, _I, _P, _I, _P]),
    "md_da3_param_inventory": (_I, [C.POINTER(MdDa3Cfg), _I, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), _F, _F]),
    "md_model_query": (_I, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "md_model_set_option": (_I, [_P, C.c_char_p, C.c_int64]),
    "md_model_enable_taps": (_I, [_P, _I]),
    "md_model_read_tap": (_I, [_P, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_int64 * 4)]),
    "md_model_enable_timing": (_I, [_P, _I]),
    "md_model_set_timing_filter": (_I, [_P, C.c_char_p]),
    "md_model_read_timing": (_I, [_P, C.POINTER(C.c_char_p), _F, C.POINTER(_I), _I, C.POINTER(_I)]),
    "md_model_read_launch_order": (_I, [_P, C.POINTER(C.c_char_p), _I, C.POINTER(_I)]),
    "md_op_rgb_to_input": (_I, [_P, _P, C.c_size_t, _I, _I, _P, _P]),
    "md_op_resize_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "md_op_pyramid_patchify": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, C.POINTER(_I), C.POINTER(_I), _P]),
    "md_op_resize_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _P]),
    "md_op_resize_output_size": (_I, [_I, _I, C.c_float, C.c_float, C.POINTER(_I), C.POINTER(_I)]),
    "md_op_split": (_I, [_P, _P, _I, _I, _I, _I, C.c_float, _P, C.POINTER(_I), _P]),
    "md_op_merge": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, C.POINTER(_I), C.POINTER(_I), _P]),
    "md_op_layernorm": (_I, [_P, _P, _P, _P, _I, _I, C.c_float, _P, _P]),
    "md_op_linear": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "md_op_linear_tile": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "md_op_attention": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "md_op_conv3x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "md_op_deconv2x2": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "md_op_conv2d_direct": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "md_op_fov_to_focal": (_I, [C.c_float, _I, _I, _F, _F]),
    "md_bench_gemm": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F]),
    "md_gemm_pick_tile": (_I, [_I, _I, _I, _I]),
    "md_bench_attention": (_I, [_P, _I, _I, _I, _I, _F]),
    "md_bench_attention_ex": (_I, [_P, _I, _I, _I, _I, C.c_float, _I, _F]),
    "md_debug_attention_asm": (_I, [_I]),
    "md_debug_attention_asm_launches": (C.c_long, []),
    "md_debug_attention_redo_units": (C.c_long, [_P, _I]),
    "md_bench_attention_qkv": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, C.POINTER(C.c_long)]),
    "md_comm_unique_id": (_I, [_P]),
    "md_comm_init_rank": (_I, [_P, _P, _I, _I, C.POINTER(_P)]),
    "md_comm_rank": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "md_comm_count": (_I, [_P, C.POINTER(_I)]),
    "md_comm_destroy": (_I, [_P]),
    "md_comm_broadcast_weights": (_I, [_P, _P, _I]),
    "md_comm_scatter_images": (_I, [_P, _P, _P, C.c_size_t, _I, _P]),
    "md_comm_gather_depth": (_I, [_P, _P, _P, C.c_size_t, _I, _P]),
    "md_depth_pro_infer_tiles_loopback": (_I, [C.POINTER(_P), _I, _I, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P]),
    "md_comm_depth_pro_infer_tiles": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P]),
    "md_param_inventory": (_I, [C.POINTER(MdDepthProCfg), _I, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), _F, _F]),
    "md_uniform_stream": (_I, [C.c_char_p, C.c_uint64, C.c_size_t, C.c_float, C.c_float, _P]),
    "md_split_geometry": (_I, [_I, _I, C.c_float, C.POINTER(_I), C.POINTER(_I)]),
    "md_feature_padding": (_I, [_I, _I, _I]),
}

_lib = None


def load() -> C.CDLL:
    """Load libmi_depth.so (built by `__graft_entry__.build()` / `make -C burn_depth_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "burn_depth_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted from the header
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != MD_OK:
        msg = load().md_last_error()
        raise MdError(code, msg.decode("utf-8", "replace") if msg else "")
