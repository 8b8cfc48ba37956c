"""ctypes binding of libladiffcodec.so (the C ABI in include/ladiffcodec.h).

There is deliberately no fallback: if the shared library is missing or does not load, `load()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libladiffcodec.so")
CSRC = os.path.join(_HERE, "csrc")

LDC_F32, LDC_BF16, LDC_BF16_W8 = 0, 1, 2
DTYPES = {"f32": LDC_F32, "bf16": LDC_BF16, "fp8": LDC_BF16_W8}
MODEL_MAIN, MODEL_COND = 0, 1
MAX_RATIOS = 8

EXPORTS = [
    "ldc_last_error", "ldc_version", "ldc_create", "ldc_destroy", "ldc_reseed", "ldc_set_option", "ldc_quantize_e4m3", "ldc_set_weight", "ldc_finalize_weights",
    "ldc_seanet_encode", "ldc_seanet_decode", "ldc_rvq_encode", "ldc_rvq_decode", "ldc_get_cond",
    "ldc_cond_upsample", "ldc_unet_forward", "ldc_p_sample", "ldc_denoise", "ldc_p_sample_loop", "ldc_infilling", "ldc_output_normalise", "ldc_decode",
    "ldc_sconv1d", "ldc_sconvtr1d", "ldc_slstm", "ldc_unet_debug_tap", "ldc_unet_step_cost", "ldc_profile_enable",
    "ldc_profile_read", "ldc_profile_read_classes", "ldc_conv_microbench", "ldc_conv_compare", "ldc_conv_compare_fp8", "ldc_ln_fold_compare", "ldc_gn_microbench", "ldc_host_stats", "ldc_stream_info", "ldc_clock_sample", "ldc_debug_raise_failure", "ldc_debug_sync_count", "ldc_xcc_census", "ldc_timeline_enable", "ldc_timeline_read", "ldc_kstamps_enable", "ldc_kstamps_reset", "ldc_kstamps_read", "ldc_packed_bytes", "ldc_pack_codes", "ldc_unpack_codes",
    "ldc_ac_build_cdf", "ldc_ac_encode", "ldc_ac_decode", "ldc_train_q_sample", "ldc_train_num_timesteps", "ldc_train_predict_x_start", "ldc_train_neg_sdsdr", "ldc_train_l1_loss", "ldc_train_block_ws_floats",
    "ldc_train_block_forward", "ldc_train_block_backward", "ldc_train_layernorm_forward", "ldc_train_layernorm_backward", "ldc_train_adam_step", "ldc_train_adam_step_dev", "ldc_train_pointwise_forward", "ldc_train_pointwise_backward", "ldc_train_linattn_ws_floats", "ldc_train_linattn_forward", "ldc_train_linattn_backward", "ldc_train_conv_forward", "ldc_train_conv_backward", "ldc_train_join", "ldc_train_side_stream", "ldc_train_upsample2", "ldc_train_activation", "ldc_train_attn_ws_floats", "ldc_train_attn_forward", "ldc_train_attn_backward", "ldc_train_convtr_forward", "ldc_train_convtr_backward", "ldc_train_maxscale", "ldc_resample_out_len", "ldc_resample",
]


class LdcConfig(C.Structure):
    _fields_ = [
        ("compute_dtype", C.c_int32), ("rep_dims", C.c_int32), ("n_filters", C.c_int32),
        ("n_residual_layers", C.c_int32), ("lstm", C.c_int32), ("n_enc_ratios", C.c_int32),
        ("enc_ratios", C.c_int32 * MAX_RATIOS), ("diff_dims", C.c_int32), ("n_upsampling_ratios", C.c_int32),
        ("upsampling_ratios", C.c_int32 * MAX_RATIOS), ("unet_scale_cond", C.c_int32), ("unet_scale_x", C.c_int32),
        ("has_cond_model", C.c_int32), ("cond_bandwidth", C.c_float), ("max_batch", C.c_int32),
        ("max_latent_len", C.c_int32), ("noise_seed", C.c_uint64), ("final_activation", C.c_int32),
        ("reserved_", C.c_int32),
    ]


# --final_activation names -> LDC_ACT_* (include/ladiffcodec.h)
FINAL_ACTIVATIONS = {None: 0, "Identity": 0, "Tanh": 1, "Sigmoid": 2, "ELU": 3, "SiLU": 4, "GELU": 5, "ReLU": 6}


# LDC_E_* (include/ladiffcodec.h)
E_INVALID, E_STATE, E_MISSING, E_HIP, E_NOMEM = -1, -2, -3, -4, -5


class LdcError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libladiffcodec error {code}: {msg}")
        self.code = code


_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode != 0:
        raise RuntimeError("building libladiffcodec.so failed")
    return LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LDC_LIB_PATH", LIB_PATH)   # tuning aid: A/B two builds of the library on one GPU box
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(or `make -C {CSRC}`). There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, i64p, fp = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p
    lib.ldc_last_error.restype = C.c_char_p
    lib.ldc_version.restype = C.c_char_p
    lib.ldc_create.argtypes = [C.POINTER(LdcConfig), i32, C.POINTER(vp)]
    lib.ldc_destroy.argtypes = [vp]
    lib.ldc_reseed.argtypes = [vp, C.c_uint64]
    lib.ldc_quantize_e4m3.argtypes = [vp, C.c_int64, vp, vp]
    lib.ldc_set_weight.argtypes = [vp, i32, C.c_char_p, vp, i64p, i32]
    lib.ldc_finalize_weights.argtypes = [vp, i32]
    lib.ldc_seanet_encode.argtypes = [vp, i32, fp, i32, i32, fp, vp]
    lib.ldc_seanet_decode.argtypes = [vp, i32, fp, i32, i32, fp, vp]
    lib.ldc_rvq_encode.argtypes = [vp, fp, i32, i32, i32, vp, fp, vp]
    lib.ldc_rvq_decode.argtypes = [vp, vp, i32, i32, i32, fp, vp]
    lib.ldc_get_cond.argtypes = [vp, fp, i32, i32, C.c_float, fp, vp, vp]
    lib.ldc_cond_upsample.argtypes = [vp, fp, i32, i32, i32, fp, vp]
    lib.ldc_unet_forward.argtypes = [vp, fp, i32, fp, i32, i32, i32, fp, vp]
    lib.ldc_p_sample.argtypes = [vp, fp, i32, fp, fp, i32, i32, i32, vp]
    lib.ldc_denoise.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.ldc_p_sample_loop.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.ldc_infilling.argtypes = [vp, fp, fp, fp, i32, fp, C.c_float, i32, i32, i32, i32, vp]
    lib.ldc_output_normalise.argtypes = [vp, fp, i32, i32, i32, vp]
    lib.ldc_decode.argtypes = [vp, fp, i32, i32, i32, fp, i32, fp, fp, fp, vp, vp]
    lib.ldc_sconv1d.argtypes = [vp, fp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, fp, vp]
    lib.ldc_sconvtr1d.argtypes = [vp, fp, i32, i32, i32, vp, vp, i32, i32, i32, i32, fp, vp]
    lib.ldc_slstm.argtypes = [vp, fp, i32, i32, i32, C.POINTER(vp), i32, fp, vp]
    lib.ldc_unet_debug_tap.argtypes = [vp, C.c_char_p, fp, C.c_int64, vp]
    lib.ldc_unet_step_cost.argtypes = [vp, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ldc_profile_enable.argtypes = [vp, i32]
    lib.ldc_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.ldc_profile_read_classes.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ldc_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.ldc_train_num_timesteps.argtypes = [vp]
    lib.ldc_train_predict_x_start.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.ldc_train_neg_sdsdr.argtypes = [vp, vp, vp, i32, C.c_int64, C.c_float, vp, vp]
    lib.ldc_host_stats.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ldc_stream_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.ldc_clock_sample.argtypes = [vp, C.POINTER(C.c_uint64), vp]
    lib.ldc_xcc_census.argtypes = [vp, vp, i32, i32, vp]
    lib.ldc_conv_microbench.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_double)]
    lib.ldc_conv_compare.argtypes = [vp] + [i32] * 13 + [C.POINTER(C.c_double)] * 3
    lib.ldc_conv_compare_fp8.argtypes = [vp] + [i32] * 10 + [C.POINTER(C.c_double)] * 3
    lib.ldc_ln_fold_compare.argtypes = [vp, i32, i32, i32, i32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ldc_timeline_enable.argtypes = [vp, i32]
    lib.ldc_timeline_read.argtypes = [vp, i32, i32, C.POINTER(C.c_uint64)]
    lib.ldc_kstamps_enable.argtypes = [vp, i32]
    lib.ldc_kstamps_reset.argtypes = [vp]
    lib.ldc_kstamps_read.argtypes = [vp, i32, i32, C.POINTER(i32), C.POINTER(C.c_uint64), C.c_char_p, i32, C.POINTER(i32)]
    lib.ldc_packed_bytes.argtypes = [i32, i32, i32]
    lib.ldc_pack_codes.argtypes = [vp, vp, i32, i32, i32, i32, vp, C.c_int64, vp]
    lib.ldc_unpack_codes.argtypes = [vp, vp, C.c_int64, i32, i32, i32, i32, vp, vp]
    lib.ldc_ac_build_cdf.argtypes = [vp, vp, i32, i32, i32, C.c_float, i32, vp, vp]
    lib.ldc_ac_encode.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, C.c_int64, vp, vp]
    lib.ldc_ac_decode.argtypes = [vp, vp, C.c_int64, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.ldc_train_q_sample.argtypes = [vp, fp, vp, fp, i32, i32, i32, fp, vp]
    lib.ldc_train_l1_loss.argtypes = [vp, fp, fp, vp, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_block_ws_floats.argtypes = [i32, i32, i32, i32, i32]
    lib.ldc_train_block_forward.argtypes = [vp, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_block_backward.argtypes = [vp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, fp, fp, fp, fp, fp, fp, fp, vp]
    lib.ldc_train_layernorm_forward.argtypes = [vp, fp, fp, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_layernorm_backward.argtypes = [vp, fp, fp, fp, fp, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_pointwise_forward.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, i32, fp, vp]
    lib.ldc_train_pointwise_backward.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, i32, fp, fp, fp, vp]
    lib.ldc_train_linattn_ws_floats.argtypes = [i32, i32, i32, i32]
    lib.ldc_train_linattn_ws_floats.restype = C.c_int64
    lib.ldc_train_linattn_forward.argtypes = [vp, fp, i32, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_linattn_backward.argtypes = [vp, fp, fp, i32, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_conv_forward.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, fp, vp]
    lib.ldc_train_conv_backward.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, fp, fp, fp, vp]
    lib.ldc_train_join.argtypes = [vp, vp]
    lib.ldc_train_side_stream.argtypes = [vp, C.POINTER(C.c_void_p)]
    lib.ldc_train_upsample2.argtypes = [vp, fp, C.c_int64, i32, i32, fp, vp]
    lib.ldc_train_activation.argtypes = [vp, fp, fp, C.c_int64, i32, fp, vp]
    lib.ldc_train_attn_ws_floats.argtypes = [i32, i32, i32]
    lib.ldc_train_attn_forward.argtypes = [vp, fp, i32, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_attn_backward.argtypes = [vp, fp, fp, i32, i32, i32, i32, fp, fp, vp]
    lib.ldc_train_convtr_forward.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, i32, fp, vp]
    lib.ldc_train_convtr_backward.argtypes = [vp, fp, fp, fp, i32, i32, i32, i32, i32, fp, fp, fp, vp]
    lib.ldc_train_maxscale.argtypes = [vp, fp, fp, i32, C.c_int64, fp, vp]
    lib.ldc_train_adam_step.argtypes = [vp, fp, fp, fp, fp, C.c_int64, i32, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    lib.ldc_train_adam_step_dev.argtypes = [vp, fp, fp, fp, fp, C.c_int64, vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    lib.ldc_resample_out_len.argtypes = [C.c_int64, i32, i32]
    lib.ldc_resample.argtypes = [vp, fp, i32, C.c_int64, i32, i32, fp, vp]
    lib.ldc_gn_microbench.argtypes = [vp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_double)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name in ("ldc_debug_sync_count", "ldc_packed_bytes", "ldc_train_block_ws_floats", "ldc_train_linattn_ws_floats", "ldc_train_attn_ws_floats", "ldc_resample_out_len"):
            fn.restype = C.c_int64
        elif name == "ldc_quantize_e4m3":
            fn.restype = None
        elif name not in ("ldc_last_error", "ldc_version"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise LdcError(rc, load().ldc_last_error().decode("utf-8", "replace"))
