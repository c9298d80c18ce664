"""ctypes binding of libnoisediff_hip.so (the C ABI in include/noisediff_hip.h).

No fallback: ``load()`` raises if the library is missing (build it with
``python -m noisediff_amd.build`` / ``__graft_entry__.build()``), and every call is
checked -- a non-zero return raises ``HipError`` with the library's message.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnoisediff_hip.so")

# enum nd_prologue / nd_act
PRO_NONE, PRO_AFFINE_SILU, PRO_AFFINE_MAP_SILU, PRO_LAYERNORM, PRO_SILU, PRO_LEAKY, PRO_LEAKY_SECOND, PRO_AFFINE_GENMAP_SILU = 0, 1, 2, 3, 4, 5, 6, 7
ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2
OBJECTIVES = {"pred_noise": 0, "pred_x0": 1, "pred_v": 2}

fptr = C.c_void_p   # device pointers travel as integers


class HipError(RuntimeError):
    pass


class Src(C.Structure):
    _fields_ = [("p0", fptr), ("p1", fptr), ("c0", C.c_int32), ("c1", C.c_int32), ("ld0", C.c_int32), ("ld1", C.c_int32),
                ("mode", C.c_int32), ("upsample", C.c_int32), ("unshuffle", C.c_int32), ("map_blocked", C.c_int32),
                ("mad", fptr), ("map", fptr), ("vec", fptr), ("gamma", fptr), ("beta", fptr), ("rowstats", fptr)]


class AdamItem(C.Structure):
    _fields_ = [("p", fptr), ("g", fptr), ("m", fptr), ("v", fptr), ("n", C.c_int64), ("step_size", C.c_float), ("bias2_sqrt", C.c_float),
                ("vec4", C.c_int32), ("reserved", C.c_int32), ("step", fptr)]


class PackItem(C.Structure):
    _fields_ = [("w", fptr), ("packed", fptr), ("cin", C.c_int32), ("cout", C.c_int32), ("transposed", C.c_int32), ("reserved", C.c_int32)]


class Conv3x3(C.Structure):
    _fields_ = [("src", Src), ("weight", fptr), ("bias", fptr), ("out", fptr), ("stats", fptr), ("slot_count", fptr),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("ldo", C.c_int32)]


class Pointwise(C.Structure):
    _fields_ = [("src", Src), ("weight", fptr), ("bias", fptr), ("out", fptr), ("res0", fptr), ("res1", fptr), ("vec", fptr),
                ("gn_t", fptr), ("gn_mad", fptr), ("B", C.c_int32), ("HW", C.c_int32), ("W", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("ldo", C.c_int32), ("ldr0", C.c_int32), ("ldr1", C.c_int32), ("ldt", C.c_int32), ("act", C.c_int32),
                ("shuffle_c", C.c_int32), ("shuffle_h", C.c_int32), ("shuffle_w", C.c_int32)]


class ChainStage(C.Structure):
    _fields_ = [("weight", fptr), ("bias", fptr), ("cin", C.c_int32), ("cout", C.c_int32), ("act", C.c_int32), ("res", C.c_int32)]


class Chain(C.Structure):
    _fields_ = [("src", Src), ("st", ChainStage * 3), ("out", fptr), ("n_stages", C.c_int32), ("B", C.c_int32), ("HW", C.c_int32),
                ("ldo", C.c_int32)]


CHAIN_RES_NONE, CHAIN_RES_INPUT, CHAIN_RES_INPUT_RAW = 0, 1, 2


class SamplerState(C.Structure):
    _fields_ = [("step", fptr), ("t_cur", fptr), ("t_next", fptr), ("coef", fptr), ("time_out", fptr), ("rng", fptr),
                ("n_steps", C.c_int32), ("B", C.c_int32)]


i32, i64, u64, vp, f32 = C.c_int, C.c_int64, C.c_uint64, C.c_void_p, C.c_float

# name -> (restype, argtypes); everything include/noisediff_hip.h declares
SIGNATURES = {
    "nd_version": (i32, []),
    "nd_last_error": (C.c_char_p, []),
    "nd_device_arch": (i32, [C.c_char_p, i32]),
    "nd_conv3x3_nhwc_f32": (i32, [C.POINTER(Conv3x3), vp]),
    "nd_conv3x3_stat_slots": (i32, [i32, i32, i32, i32]),
    "nd_conv3x3_tiling_id": (i32, [i32, i32, i32, i32]),
    "nd_pack_conv3x3_weight_floats": (i64, [i32, i32]),
    "nd_pack_conv3x3_weight": (i32, [vp, vp, i32, i32, vp]),
    "nd_pack_conv3x3_weight_dgrad": (i32, [vp, vp, i32, i32, vp]),
    "nd_conv3x3_wino_nhwc_f32": (i32, [C.POINTER(Conv3x3), vp]),
    "nd_conv3x3_wino2_nhwc_f32": (i32, [C.POINTER(Conv3x3), vp]),
    "nd_conv3x3_wino_stat_slots": (i32, [i32, i32]),
    "nd_conv3x3_wino4_stat_slots": (i32, [i32, i32]),
    "nd_pack_conv3x3_wino_weight_floats": (i64, [i32, i32]),
    "nd_pack_conv3x3_wino_weight": (i32, [vp, vp, i32, i32, vp]),
    "nd_pack_conv3x3_wino_weight_dgrad": (i32, [vp, vp, i32, i32, vp]),
    "nd_conv3x3_wino4_nhwc_f32": (i32, [C.POINTER(Conv3x3), vp]),
    "nd_conv3x3_wino4_16_nhwc_f32": (i32, [C.POINTER(Conv3x3), vp]),
    "nd_pack_conv3x3_wino4_weight_floats": (i64, [i32, i32]),
    "nd_pack_conv3x3_wino4_weight": (i32, [vp, vp, i32, i32, vp]),
    "nd_pack_conv3x3_wino4_weight_dgrad": (i32, [vp, vp, i32, i32, vp]),
    "nd_conv3x3_wino4_splitk_plan": (i32, [i32, i32, i32, i32, i32]),
    "nd_conv3x3_wino4_splitk_workspace_floats": (i64, [i32, i32, i32, i32, i32]),
    "nd_conv3x3_wino4_splitk_nhwc_f32": (i32, [vp, vp, i32, vp]),
    "nd_conv3x3_wino4_16_splitk_plan": (i32, [i32, i32, i32, i32]),
    "nd_conv3x3_wino4_16_splitk_nhwc_f32": (i32, [vp, vp, i32, vp]),
    "nd_conv3x3_wgrad_workspace_floats": (i64, [i32, i32, i32, i32, i32]),
    "nd_conv3x3_wgrad_nhwc_f32": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nd_groupnorm_train_workspace_floats": (i64, [i32, i32, i32]),
    "nd_linear_wgrad_workspace_floats": (i64, [i64, i32, i32]),
    "nd_groupnorm_silu_train_workspace_floats": (i64, [i32, i32, i32]),
    "nd_groupnorm_silu_train_forward_f32": (i32, [vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, C.c_float, vp]),
    "nd_token_sum_workspace_floats": (i64, [i32, i32, i32]),
    "nd_token_sum_f32": (i32, [vp, i32, vp, vp, i32, i32, i32, vp]),
    "nd_modulate_silu_forward_f32": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, vp]),
    "nd_modulate_silu_backward_f32": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i64, i32, vp]),
    "nd_groupnorm_silu_train_backward_f32": (i32, [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nd_layernorm_train_workspace_floats": (i64, [i64, i32]),
    "nd_layernorm_train_forward_f32": (i32, [vp, i32, vp, vp, vp, i32, vp, i64, i32, C.c_float, vp]),
    "nd_layernorm_train_backward_f32": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp, i64, i32, vp]),
    "nd_linear_wgrad_f32": (i32, [vp, i32, vp, i32, vp, vp, vp, i64, i32, i32, vp]),
    "nd_groupnorm_train_forward_f32": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, C.c_float, vp]),
    "nd_groupnorm_train_backward_f32": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nd_pointwise_gemm_nhwc_f32": (i32, [C.POINTER(Pointwise), vp]),
    "nd_pointwise_chain_nhwc_f32": (i32, [C.POINTER(Chain), vp]),
    "nd_pointwise_chain_supported": (i32, [i32, i32, i32, i32]),
    "nd_pack_chain_weight_floats": (i64, [i32, i32, i32]),
    "nd_pack_chain_weight": (i32, [vp, vp, i32, i32, i32, vp]),
    "nd_pointwise_chain_split_nhwc_f32": (i32, [C.POINTER(Chain), vp]),
    "nd_pack_chain_weight_split_floats": (i64, [i32, i32, i32]),
    "nd_pack_chain_weight_split": (i32, [vp, vp, i32, i32, i32, vp]),
    "nd_pointwise_gemm_split_nhwc_f32": (i32, [C.POINTER(Pointwise), vp]),
    "nd_pointwise_gemm_split_takes": (i32, [C.POINTER(Pointwise)]),
    "nd_pack_pointwise_weight_split_floats": (i64, [i32, i32]),
    "nd_pack_pointwise_weight_split": (i32, [vp, vp, i32, i32, vp]),
    "nd_pack_pointwise_weight_floats": (i64, [i32, i32]),
    "nd_pack_pointwise_weight": (i32, [vp, vp, i32, i32, i32, vp]),
    "nd_pack_pointwise_weight_t": (i32, [vp, vp, i32, i32, vp]),
    "nd_pack_pointwise_weights_batch": (i32, [vp, i32, vp]),
    "nd_pack_conv3x3_wino4_weights_batch": (i32, [vp, i32, vp]),
    "nd_conv3x3_wgrad_cat_nhwc_f32": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nd_conv3x3_wgrad_form": (i32, [i32]),
    "nd_conv7x7_c4_wgrad_workspace_floats": (i64, [i32, i32, i32, i32]),
    "nd_conv7x7_c4_wgrad_f32": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nd_adam_chunk_elements": (i32, []),
    "nd_adam_step_f32": (i32, [vp, i32, vp, i32, f32, f32, f32, f32, vp]),
    "nd_adam_step_capturable_f32": (i32, [vp, i32, vp, i32, f32, f32, f32, f32, f32, vp]),
    "nd_groupnorm_finalize_f32": (i32, [vp, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, f32, vp]),
    "nd_groupnorm_finalize_train_f32": (i32, [vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, f32, vp]),
    "nd_layernorm_stats_f32": (i32, [vp, i32, vp, vp, i32, i32, i32, f32, vp]),
    "nd_affine_silu_add_f32": (i32, [vp, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "nd_linear_rows_f32": (i32, [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "nd_sinusoidal_time_emb_f32": (i32, [vp, vp, vp, i32, i32, vp]),
    "nd_cond_step_lds_bytes": (i64, [i32, i32]),
    "nd_cond_step_f32": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nd_cond_table_build_f32": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "nd_cond_step_table_f32": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "nd_cond_step_ptable_f32": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp]),
    "nd_embedding_rows_f32": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "nd_conv7x7_c4_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nd_pack_conv7x7_weight": (i32, [vp, vp, i32, vp]),
    "nd_conv7x7_c4_split_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nd_pack_conv7x7_weight_split_floats": (i64, [i32]),
    "nd_pack_conv7x7_weight_split": (i32, [vp, vp, i32, vp]),
    "nd_pos_enc_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nd_maxpool2x2_nhwc_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nd_nchw_to_nhwc_pad_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "nd_nchw_to_nhwc_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nd_nhwc_to_nchw_f32": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nd_sampler_begin_step": (i32, [C.POINTER(SamplerState), vp]),
    "nd_sampler_advance": (i32, [C.POINTER(SamplerState), vp]),
    "nd_sampler_step_ddpm_f32": (i32, [vp, vp, vp, i64, C.POINTER(SamplerState), i32, u64, i64, i32, i32, i32, vp]),
    "nd_sampler_step_ddim_f32": (i32, [vp, vp, vp, i64, C.POINTER(SamplerState), i32, u64, i64, i32, i32, i32, vp]),
    "nd_philox_normal_f32": (i32, [vp, u64, i64, i32, i32, i32, i32, vp]),
    "nd_attention_mfma_f32": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "nd_linear_attention_workspace_floats": (i64, [i32, i32, i32]),
    "nd_linear_attention_f32": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "nd_rmsnorm_nhwc_f32": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, vp]),
    "nd_rmsnorm_add_nhwc_f32": (i32, [vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, vp]),
    "nd_stream_create": (i32, [C.POINTER(vp)]),
    "nd_stream_destroy": (i32, [vp]),
    "nd_stream_sync": (i32, [vp]),
    "nd_graph_begin": (i32, [vp]),
    "nd_graph_end": (i32, [vp, C.POINTER(vp)]),
    "nd_graph_launch": (i32, [vp, vp]),
    "nd_graph_destroy": (i32, [vp]),
    "nd_event_create": (i32, [C.POINTER(vp)]),
    "nd_event_record": (i32, [vp, vp]),
    "nd_event_elapsed_ms": (i32, [vp, vp, C.POINTER(f32)]),
    "nd_event_destroy": (i32, [vp]),
    "nd_event_create_untimed": (i32, [C.POINTER(vp)]),
    "nd_stream_wait_event": (i32, [vp, vp]),
    "nd_stream_device": (i32, [vp]),
}

_UNCHECKED = {"nd_version", "nd_last_error", "nd_stream_device", "nd_conv3x3_wgrad_form", "nd_adam_chunk_elements", "nd_conv7x7_c4_wgrad_workspace_floats", "nd_conv3x3_stat_slots", "nd_conv3x3_tiling_id", "nd_pack_conv3x3_weight_floats",
              "nd_pack_pointwise_weight_floats", "nd_linear_attention_workspace_floats", "nd_conv3x3_wino_stat_slots", "nd_conv3x3_wino4_stat_slots",
              "nd_pack_conv3x3_wino_weight_floats", "nd_pack_conv3x3_wino4_weight_floats", "nd_conv3x3_wino4_splitk_plan", "nd_conv3x3_wino4_16_splitk_plan", "nd_conv3x3_wino4_splitk_workspace_floats", "nd_token_sum_workspace_floats", "nd_cond_step_lds_bytes", "nd_conv3x3_wgrad_workspace_floats",
              "nd_groupnorm_train_workspace_floats", "nd_linear_wgrad_workspace_floats",
              "nd_layernorm_train_workspace_floats", "nd_groupnorm_silu_train_workspace_floats"}

_lib: Optional[C.CDLL] = None


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the library and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: it ships its own libamdhip64; loading ours before torch's puts two HIP runtimes in the process
    # and the second one to initialise reports "no ROCm-capable device"
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise HipError(f"{path} is missing: the HIP library is the product and there is no fallback. "
                       "Build it with `python -m noisediff_amd.build`.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().nd_last_error().decode(errors="replace")
        raise HipError(f"{what or 'libnoisediff_hip'} failed with code {code}: {msg}")


def call(name: str, *args):
    """Checked call: raises HipError on a non-zero status."""
    r = getattr(load(), name)(*args)
    if name not in _UNCHECKED:
        check(r, name)
    return r


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
