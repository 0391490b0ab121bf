"""ctypes binding of libmcvd_hip.so (C ABI: include/mcvd_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing this
module raises.  Build it with `python mcvd_pytorch_amd/csrc/build.py` (or `__graft_entry__.build()`).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- MUST precede the CDLL: libmcvd_hip.so then binds to the HIP runtime torch already loaded
#                              (two libamdhip64 copies in one process do not both see the device)

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCVD_LIB_PATH: diagnostics only (tests/gpu_diag.py loads libmcvd_hip_diag.so, the -DMCVD_DIAG build with the ablation kernels)
LIB_PATH = os.environ.get("MCVD_LIB_PATH") or os.path.join(_HERE, "libmcvd_hip.so")

MAX_LEVELS = 8


class UNetDesc(C.Structure):
    _fields_ = [("image_size", C.c_int32), ("channels", C.c_int32), ("num_frames", C.c_int32),
                ("num_frames_cond", C.c_int32), ("ngf", C.c_int32), ("n_levels", C.c_int32),
                ("ch_mult", C.c_int32 * MAX_LEVELS), ("num_res_blocks", C.c_int32), ("n_attn", C.c_int32),
                ("attn_resolutions", C.c_int32 * MAX_LEVELS), ("n_head_channels", C.c_int32), ("spade", C.c_int32),
                ("spade_dim", C.c_int32), ("num_classes", C.c_int32), ("sigma_dist", C.c_int32),
                ("sigma_begin", C.c_float), ("sigma_end", C.c_float), ("cond_emb", C.c_int32), ("noise_in_cond", C.c_int32),
                ("gamma", C.c_int32)]


SAMPLER_DDPM, SAMPLER_DDIM = 0, 1
FLAG_DENOISE, FLAG_CLIP_BEFORE, FLAG_JUST_BETA, FLAG_GAMMA = 1, 2, 4, 8

_vp, _i, _f, _i64, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint64
_PROTOS = {
    "mcvd_version": (C.c_char_p, []),
    "mcvd_last_error": (C.c_char_p, [_vp]),
    "mcvd_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "mcvd_ctx_destroy": (None, [_vp]),
    "mcvd_ctx_device_shared": (_i, [_vp]),
    "mcvd_ctx_set_stream": (_i, [_vp, _vp]),
    "mcvd_ctx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "mcvd_ctx_check_range": (_i, [_vp]),
    "mcvd_ctx_clear_range": (_i, [_vp]),
    "mcvd_ctx_selftest": (_i, [_vp]),
    "mcvd_ctx_set_debug_buffer": (_i, [_vp, _vp]),
    "mcvd_model_create": (_i, [_vp, C.POINTER(UNetDesc), C.POINTER(_vp)]),
    "mcvd_model_destroy": (None, [_vp]),
    "mcvd_model_num_params": (_i, [_vp]),
    "mcvd_model_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i), C.POINTER(_i64)]),
    "mcvd_model_set_param": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i, _i]),
    "mcvd_model_blob_floats": (_i, [_vp, C.POINTER(_i64)]),
    "mcvd_model_export_blob": (_i, [_vp, _vp]),
    "mcvd_model_import_blob": (_i, [_vp, _vp]),
    "mcvd_model_broadcast_params": (_i, [_vp, _vp, _i]),
    "mcvd_model_finalize": (_i, [_vp]),
    "mcvd_model_get_schedule": (_i, [_vp, _vp, _vp, _vp, _i]),
    "mcvd_model_set_schedule": (_i, [_vp, _vp, _vp, _vp, _i]),
    "mcvd_model_set_temb_freqs": (_i, [_vp, _vp, _i]),
    "mcvd_unet_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    "mcvd_unet_forward_ft": (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    "mcvd_unet_forward_masked": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "mcvd_model_set_cond_noise": (_i, [_vp, _vp, _u64, _u64, _u64]),
    "mcvd_model_set_gamma_tables": (_i, [_vp, _vp, _vp, _i]),
    "mcvd_model_prepare_cond": (_i, [_vp, _vp, _i]),
    "mcvd_model_invalidate_cond": (_i, [_vp]),
    "mcvd_model_num_launches": (_i, [_vp, _i]),
    "mcvd_model_get_tuning": (_i, [_vp, _i, _vp, _vp, _i]),
    "mcvd_model_set_tuning": (_i, [_vp, _i, _vp, _vp, _i]),
    "mcvd_model_graph_stats": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "mcvd_last_conv_kernel": (_i, []),
    "mcvd_last_conv_stats_np": (_i, []),
    "mcvd_ctx_set_stats_buffer": (_i, [_vp, _vp]),
    "mcvd_ctx_set_spade_inputs": (_i, [_vp, _vp, _vp]),
    "mcvd_op_gn_finalize": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _f, _i, _vp, _vp, _i, _i, _vp, _i, _i]),
    "mcvd_model_profile_read": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "mcvd_model_op_info": (_i, [_vp, _i, C.POINTER(_i)]),
    "mcvd_model_op_kernel": (_i, [_vp, _i]),
    "mcvd_model_fused_launches": (C.c_long, [_vp, _i]),
    "mcvd_model_module_output": (_i, [_vp, _i, _i, _vp, _i64, C.POINTER(_i), C.POINTER(_i)]),
    "mcvd_sampler_run": (_i, [_vp, _i, _vp, _vp, _vp, _u64, _u64, _i, _i, C.c_double, _i]),
    "mcvd_fpndm_run": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "mcvd_sampler_update": (_i, [_vp, _i, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _i64]),
    "mcvd_randn": (_i, [_vp, _vp, _u64, _u64, _u64, _i, _i64]),
    "mcvd_pack_frames_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "mcvd_gamma_noise": (_i, [_vp, _vp, _vp, _f, _f, _f, _f, _u64, _u64, _u64, _i, _i64]),
    "mcvd_lincomb": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _i64]),
    "mcvd_pndm_transfer": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _i64]),
    "mcvd_upfirdn2d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i]),
    "mcvd_op_conv2d": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _f, _vp, _i, _i, _i]),
    "mcvd_op_gn_coef": (_i, [_vp, _vp, _i, _vp, _i, _i, _f, _i, _vp, _vp, _i, _i, _vp, _i, _i]),
    "mcvd_op_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "mcvd_op_fir2": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i]),
}

EXPORTS = tuple(_PROTOS.keys())


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python mcvd_pytorch_amd/csrc/build.py` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    return (lib.mcvd_last_error(None) or b"").decode()


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libmcvd_hip: {what} failed (code {rc}): {last_error()}")
