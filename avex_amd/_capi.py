"""ctypes binding of include/avexhip.h (the C-ABI shared library built by avex_amd.build).

The product path has NO CPU fallback: if the library is missing or no GPU is visible the
callers raise ``AvexHipError`` loudly instead of silently computing somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVEX_AMD_LIB") or os.path.join(HERE, "lib", "libavexhip.so")   # AVEX_AMD_LIB: A/B builds of the same ABI

F16, BF16 = 0, 1
DTYPE_NAMES = {"f16": F16, "fp16": F16, "float16": F16, "bf16": BF16, "bfloat16": BF16}


class AvexHipError(RuntimeError):
    """Raised for any failure of the HIP extension (missing library, no GPU, bad arguments)."""


class FbankConfig(C.Structure):
    _fields_ = [("win_length", C.c_int32), ("hop_length", C.c_int32), ("n_mels", C.c_int32),
                ("input_scale", C.c_float), ("preemph", C.c_float), ("remove_dc", C.c_int32),
                ("log_floor", C.c_float), ("norm_mean", C.c_float), ("norm_div", C.c_float)]


class MelspecConfig(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("hop_length", C.c_int32), ("win_length", C.c_int32), ("n_mels", C.c_int32),
                ("center", C.c_int32), ("normalize", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("ldw", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("bias", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int64), ("alpha", C.c_float),
                ("resid_half", C.c_void_p), ("ldrh", C.c_int64), ("gelu", C.c_int32), ("out_f32", C.c_void_p), ("ldo", C.c_int64),
                ("out_half", C.c_void_p), ("ldh", C.c_int64), ("out_raw", C.c_void_p), ("ldraw", C.c_int64),
                ("variant", C.c_int32),
                ("ln_rows", C.c_void_p), ("ln_s", C.c_void_p),
                ("lnr_y", C.c_void_p), ("ldy", C.c_int64), ("lnr_rows", C.c_void_p),
                ("lnr_gamma", C.c_void_p), ("lnr_beta", C.c_void_p), ("stats_out", C.c_void_p),
                ("overflow_count", C.c_void_p), ("pool_part", C.c_void_p), ("pool_rows", C.c_int32), ("pool_mode", C.c_int32),
                ("splitk_ws", C.c_void_p), ("splitk_bytes", C.c_size_t), ("rows_out", C.c_void_p), ("rows_eps", C.c_float)]


class BeatsConfig(C.Structure):
    _fields_ = [("input_patch_size", C.c_int32), ("embed_dim", C.c_int32), ("encoder_layers", C.c_int32),
                ("encoder_embed_dim", C.c_int32), ("encoder_ffn_embed_dim", C.c_int32),
                ("encoder_attention_heads", C.c_int32), ("conv_pos", C.c_int32), ("conv_pos_groups", C.c_int32),
                ("num_buckets", C.c_int32), ("max_distance", C.c_int32), ("gru_rel_pos", C.c_int32),
                ("deep_norm", C.c_int32), ("num_mel_bins", C.c_int32), ("sample_frequency", C.c_float),
                ("frame_length_ms", C.c_float), ("frame_shift_ms", C.c_float), ("fbank_mean", C.c_float),
                ("fbank_std", C.c_float), ("operand_dtype", C.c_int32), ("max_chunk_clips", C.c_int32),
                ("residual_dtype", C.c_int32), ("layer_norm_first", C.c_int32), ("activation_fn", C.c_int32),
                ("conv_bias", C.c_int32), ("hidden_shift", C.c_int32)]


# activation_fn codes of BeatsConfig (AVEXHIP_FFN_* in include/avexhip.h), keyed by get_activation_fn's names (modules.py:203-237)
FFN_CODES = {"gelu": 0, "relu": 1, "gelu_accurate": 2, "gelu_fast": 2, "tanh": 3, "linear": 4, "glu": 5}


class EatConfig(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("num_heads", C.c_int32), ("depth", C.c_int32), ("ffn_dim", C.c_int32), ("patch_size", C.c_int32),
                ("target_length", C.c_int32), ("n_mels", C.c_int32), ("norm_eps", C.c_float), ("norm_mean", C.c_float), ("norm_std", C.c_float),
                ("operand_dtype", C.c_int32), ("max_chunk_clips", C.c_int32), ("residual_dtype", C.c_int32)]


class StackConfig(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("ffn_dim", C.c_int32), ("norm_eps", C.c_float),
                ("activation", C.c_int32), ("operand_dtype", C.c_int32), ("max_chunk_clips", C.c_int32), ("residual_dtype", C.c_int32)]


class AvesConfig(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("ffn_dim", C.c_int32),
                ("pos_conv_kernel", C.c_int32), ("pos_conv_groups", C.c_int32), ("n_conv_layers", C.c_int32),
                ("conv_kernel", C.c_int32 * 8), ("conv_stride", C.c_int32 * 8),
                ("operand_dtype", C.c_int32), ("max_chunk_clips", C.c_int32), ("residual_dtype", C.c_int32)]


class EffnetConfig(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("stage", (C.c_int32 * 6) * 8), ("stem_channels", C.c_int32), ("head_channels", C.c_int32),
                ("bn_eps", C.c_float), ("operand_dtype", C.c_int32), ("max_chunk_clips", C.c_int32)]


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


# every symbol include/avexhip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "avexhip_last_error": (C.c_char_p, []),
    "avexhip_abi_version": (C.c_int, []),
    "avexhip_device_count": (C.c_int, []),
    "avexhip_fbank_plan_create": (_P, [C.POINTER(FbankConfig), _P, _P]),
    "avexhip_fbank_plan_destroy": (None, [_P]),
    "avexhip_fbank_num_frames": (C.c_int, [_P, C.c_int64]),
    "avexhip_fbank_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, _P]),
    "avexhip_melspec_plan_create": (_P, [C.POINTER(MelspecConfig), _P, _P]),
    "avexhip_melspec_plan_destroy": (None, [_P]),
    "avexhip_melspec_num_frames": (C.c_int, [_P, C.c_int64]),
    "avexhip_melspec_num_bins": (C.c_int, [_P]),
    "avexhip_melspec_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, _P, _P]),
    "avexhip_effnet_stem": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P, _P, C.c_int, _P]),
    "avexhip_effnet_dwconv_part_bytes": (C.c_size_t, [C.c_int] * 6),
    "avexhip_effnet_dwconv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_size_t, C.c_int, _P]),
    "avexhip_effnet_se": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "avexhip_resample_plan_create": (_P, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]),
    "avexhip_resample_interp_plan_create": (_P, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "avexhip_resample_plan_destroy": (None, [_P]),
    "avexhip_resample_out_length": (C.c_int64, [_P, C.c_int64]),
    "avexhip_resample_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_int64, _P]),
    "avexhip_pcm_to_mono_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P]),
    "avexhip_flac_open": (_P, [_P, C.c_size_t]),
    "avexhip_flac_close": (None, [_P]),
    "avexhip_flac_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), _P]),
    "avexhip_flac_decode_i32": (C.c_int, [_P, _P, C.c_int, _P]),
    "avexhip_wavconv0_frames": (C.c_int, [C.c_int64]),
    "avexhip_wavconv0_stats_floats": (C.c_int64, [C.c_int, C.c_int64]),
    "avexhip_wavconv0": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, _P, _P, C.c_float, _P, _P, C.c_int, C.c_int, _P]),
    "avexhip_seq_interp_linear": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "avexhip_layer_mix": (C.c_int, [C.POINTER(_P), C.c_int, _P, C.c_int64, _P, _P]),
    "avexhip_dense_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "avexhip_mha_f32": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "avexhip_stack_create": (_P, [_P, _P, C.c_int]),
    "avexhip_stack_destroy": (None, [_P]),
    "avexhip_stack_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "avexhip_stack_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "avexhip_stack_overflow_count": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, C.c_int]),
    "avexhip_lstm_layer": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "avexhip_lstm_layer_pair": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "avexhip_clip_mean": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, _P]),
    "avexhip_fbank_forward_padded": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_int, _P, _P]),
    "avexhip_fbank_forward_patches": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_int, C.c_int, _P, C.c_int, _P]),
    "avexhip_token_embed_ln": (C.c_int, [_P, _P, _P, _P, _P, C.c_float, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "avexhip_cast_f32_to_half": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "avexhip_cast_half_to_f32": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "avexhip_gemm": (C.c_int, [C.POINTER(GemmArgs), C.c_int, _P]),
    "avexhip_pool_reduce": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "avexhip_pool_reduce_mode": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, C.c_int, _P]),
    "avexhip_ln_rowstats": (C.c_int, [_P, C.c_int, C.c_int, C.c_float, _P, _P]),
    "avexhip_layernorm": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_float, C.c_int, C.c_int, _P, C.c_int64, _P,
                                    C.c_int64, C.c_int, _P]),
    "avexhip_attention": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int, _P]),
    "avexhip_attention_hd": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "avexhip_posconv_pack": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P]),
    "avexhip_posconv": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "avexhip_mean_pool": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "avexhip_rel_bucket": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "avexhip_beats_create": (_P, [C.POINTER(BeatsConfig), C.POINTER(Tensor), C.c_int]),
    "avexhip_beats_destroy": (None, [_P]),
    "avexhip_beats_num_tokens": (C.c_int, [_P, C.c_int64]),
    "avexhip_beats_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int64]),
    "avexhip_beats_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_uint32, C.POINTER(_P),
                                        C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "avexhip_beats_forward_fbank": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_uint32, C.POINTER(_P), C.c_int,
                                              _P, _P, _P, C.c_size_t, _P]),
    "avexhip_beats_graph_capture": (_P, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_uint32, C.POINTER(_P),
                                         C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "avexhip_beats_graph_launch": (C.c_int, [_P, _P]),
    "avexhip_beats_graph_nodes": (C.c_int, [_P]),
    "avexhip_beats_graph_destroy": (None, [_P]),
    "avexhip_beats_overflow_count": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, C.c_int]),
    "avexhip_beats_overflow_reset": (C.c_int, [_P, _P]),
    "avexhip_beats_set_profiling": (C.c_int, [_P, C.c_int]),
    "avexhip_beats_last_profile": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float)),
                                             C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int)]),
    "avexhip_eat_create": (_P, [C.POINTER(EatConfig), C.POINTER(Tensor), C.c_int]),
    "avexhip_eat_destroy": (None, [_P]),
    "avexhip_eat_num_tokens": (C.c_int, [_P]),
    "avexhip_eat_workspace_bytes": (C.c_size_t, [_P, C.c_int]),
    "avexhip_eat_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_uint32, C.POINTER(_P), C.c_int, _P, _P, C.c_int, _P, C.c_size_t, _P]),
    "avexhip_eat_overflow_count": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, C.c_int]),
    "avexhip_eat_set_profiling": (C.c_int, [_P, C.c_int]),
    "avexhip_eat_last_profile": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_double)),
                                           C.POINTER(C.c_int)]),
    "avexhip_aves_create": (_P, [C.POINTER(AvesConfig), C.POINTER(Tensor), C.c_int]),
    "avexhip_aves_destroy": (None, [_P]),
    "avexhip_aves_num_tokens": (C.c_int, [_P, C.c_int64]),
    "avexhip_aves_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int64]),
    "avexhip_aves_forward": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int64, _P, C.c_uint32, C.POINTER(_P), C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "avexhip_aves_overflow_count": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, C.c_int]),
    "avexhip_aves_set_profiling": (C.c_int, [_P, C.c_int]),
    "avexhip_aves_last_profile": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_double)),
                                            C.POINTER(C.c_int)]),
    "avexhip_effnet_create": (_P, [C.POINTER(EffnetConfig), C.POINTER(Tensor), C.c_int]),
    "avexhip_effnet_destroy": (None, [_P]),
    "avexhip_effnet_num_taps": (C.c_int, [_P]),
    "avexhip_effnet_tap_shape": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "avexhip_effnet_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    "avexhip_effnet_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(_P), _P, _P, _P, C.c_size_t, _P]),
    "avexhip_effnet_overflow_count": (C.c_int, [_P, C.POINTER(C.c_uint32), _P, C.c_int]),
    "avexhip_effnet_set_profiling": (C.c_int, [_P, C.c_int]),
    "avexhip_effnet_last_profile": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_double)),
                                              C.POINTER(C.c_int)]),
}

# exported by the diagnostic build only (-DAVEX_DIAG; AVEX_AMD_DIAG=1 python -m avex_amd.build, then AVEX_AMD_LIB=.../libavexhip_diag.so)
DIAG_SYMBOLS = {
    "avexhip_debug_lds_canary": (C.c_int, [C.c_int, C.c_int, _P, _P]),
    "avexhip_debug_gemm_stamps": (C.c_int, [C.c_int, _P, C.c_int]),
    "avexhip_debug_gemm_clocks": (C.c_int, [_P, C.c_int]),
    "avexhip_debug_gemm_kclocks": (C.c_int, [_P, C.c_int]),
    "avexhip_debug_att_stamps": (C.c_int, [_P, C.c_int]),       # -DATT_STAMPS=1 builds only
}

_lib: Optional[C.CDLL] = None


def header_abi_version() -> int:
    """AVEXHIP_ABI_VERSION as include/avexhip.h declares it (the one place the number is written)."""
    import re
    hdr = os.path.join(os.path.dirname(HERE), "include", "avexhip.h")
    with open(hdr) as f:
        m = re.search(r"^#define\s+AVEXHIP_ABI_VERSION\s+(\d+)", f.read(), re.M)
    if not m:
        raise AvexHipError(f"{hdr} does not define AVEXHIP_ABI_VERSION")
    return int(m.group(1))


def lib() -> C.CDLL:
    """Load libavexhip.so (once).  Raises AvexHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AvexHipError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -m avex_amd.build` "
            "(needs hipcc). There is no CPU fallback for this path."
        )
    try:
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the box
        raise AvexHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise AvexHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DIAG_SYMBOLS.items():
        fn = getattr(handle, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = handle
    return handle


def last_error() -> str:
    msg = lib().avexhip_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise AvexHipError(f"{what} failed (code {rc}): {last_error()}")


def require_gpu() -> None:
    if lib().avexhip_device_count() <= 0:
        raise AvexHipError("no HIP device visible: the avex_amd compute path runs on MI355X only (no CPU fallback)")


def dtype_code(name) -> int:
    if isinstance(name, int):
        return name
    try:
        return DTYPE_NAMES[str(name).lower()]
    except KeyError as e:
        raise ValueError(f"unknown operand dtype {name!r} (use 'f16' or 'bf16')") from e
