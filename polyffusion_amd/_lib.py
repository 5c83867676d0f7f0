"""ctypes binding of libpfhip.so (the C ABI in include/pfhip.h).

There is no fallback: if the shared library is missing or a call fails, a RuntimeError
is raised.  ``load()`` itself needs no GPU (the library dlopens on a CPU-only box, which
the ``-m "not gpu"`` tests use to check the exported symbols); launching needs one.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# The library exists in two builds of the same sources (csrc/pf_internal.h): libpfhip.so splits operands into bf16 pieces ("bf16x3"),
# libpfhip_f16.so (`python -m polyffusion_amd.build --variant=f16`) into fp16 pieces ("f16x3": fp32-class error, ~3 % slower, fp16's range).
# load() returns the process default - libpfhip.so unless PF_X3=f16 is set, which makes everything named "bf16x3" run as f16x3 (how the
# whole GPU suite is run against the fp16 build); load("f16") returns the fp16 build explicitly (UNetModel.set_precision("f16x3")).
X3_VARIANT = {"": "", "bf16": "", "f16": "f16"}.get(os.environ.get("PF_X3", ""), None)
if X3_VARIANT is None:
    raise RuntimeError(f"PF_X3={os.environ['PF_X3']!r}: expected 'bf16' (default) or 'f16'")


def lib_path(variant: str = "") -> str:
    return os.path.join(HERE, f"libpfhip_{variant}.so" if variant else "libpfhip.so")


LIB_PATH = lib_path(X3_VARIANT)
X3_WEIGHT_SCALE = 256.0 if X3_VARIANT == "f16" else 1.0     # what the default library's split packing multiplies weights by (PF_X3_WS)


def x3_torch_dtype(variant: Optional[str] = None):
    """Element type of the hi / lo planes the split-precision kernels exchange (tests decode planes with it)."""
    import torch
    return torch.float16 if (X3_VARIANT if variant is None else variant) == "f16" else torch.bfloat16


c_float_p = C.POINTER(C.c_float)
c_i64_p = C.POINTER(C.c_int64)


class UNetCfg(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("channels", C.c_int32), ("n_res_blocks", C.c_int32),
        ("n_attention_levels", C.c_int32), ("attention_levels", C.c_int32 * 8),
        ("n_levels", C.c_int32), ("channel_multipliers", C.c_int32 * 8),
        ("n_heads", C.c_int32), ("tf_layers", C.c_int32), ("d_cond", C.c_int32),
        ("img_h", C.c_int32), ("img_w", C.c_int32),
    ]


class DdpmCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("c_recip", "c_recipm1", "c_x0", "c_xt", "sigma", "sqrt_ab", "sqrt_1mab")]


class DdimCoef(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("s1m", "sqrt_a", "sqrt_aprev", "dir_coef", "sigma", "q_sqrt_a", "q_s1m")]


class UNetPrepared(C.Structure):
    """pf_unet_prepared (include/pfhip.h): the step-invariant prefix a sampler computes once per loop."""
    _fields_ = [("time_table", C.c_void_p), ("n_time_rows", C.c_int32), ("cross_bias", C.c_void_p)]


OPT_MLP_FUSED, OPT_ATTN_WIDE, OPT_CONV_T16, OPT_CONV_PP, OPT_CONV_WINO = 0, 1, 2, 3, 4
OPT_COUNT = 5              # PF_OPT_COUNT
OPT_AUTO, OPT_OFF, OPT_ON = -1, 0, 1


class ConvArgs(C.Structure):
    _fields_ = [
        ("x0", C.c_void_p), ("c0", C.c_int32), ("x1", C.c_void_p), ("c1", C.c_int32),
        ("batch", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("ks", C.c_int32), ("stride", C.c_int32), ("ups", C.c_int32),
        ("w", C.c_void_p), ("n", C.c_int32),
        ("prologue", C.c_int32), ("sc", C.c_void_p), ("sh", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
        ("bias", C.c_void_p), ("sbias", C.c_void_p), ("ld_sbias", C.c_int32), ("res", C.c_void_p), ("ld_res", C.c_int32),
        ("geglu", C.c_int32),
        ("out", C.c_void_p), ("ld_out", C.c_int32),
        ("precision", C.c_int32),
        ("stats_out", C.c_void_p),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_size_t),
        ("a_planes", C.c_int32), ("out_planes", C.c_void_p),
        ("qkv_planes", C.c_void_p),
        ("ups_fold", C.c_int32),
        ("skip_x0", C.c_void_p), ("skip_c0", C.c_int32), ("skip_x1", C.c_void_p), ("skip_c1", C.c_int32),
        ("skip_w", C.c_void_p), ("skip_bias", C.c_void_p),
        ("gn_stats0", C.c_void_p), ("gn_tiles0", C.c_int32), ("gn_stats1", C.c_void_p), ("gn_tiles1", C.c_int32),
        ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_eps", C.c_float), ("gn_groups", C.c_int32),
        ("sbias_rows", C.c_void_p), ("sbias_nrows", C.c_int32), ("no_t16", C.c_int32), ("no_pp", C.c_int32),
        ("force_tile", C.c_int32), ("force_ksplit", C.c_int32), ("x1_bmod", C.c_int32),
        ("w_wino", C.c_void_p), ("wino", C.c_int32), ("absmax_slot", C.c_void_p),
    ]


# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against include/pfhip.h
SIGNATURES = {
    "pf_version": (C.c_int, []),
    "pf_last_error": (C.c_char_p, []),
    "pf_unet_create": (C.c_int, [C.POINTER(UNetCfg), C.POINTER(C.c_void_p)]),
    "pf_unet_destroy": (None, [C.c_void_p]),
    "pf_unet_weight_bytes": (C.c_size_t, [C.c_void_p]),
    "pf_unet_n_params": (C.c_int, [C.c_void_p]),
    "pf_unet_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, c_i64_p, C.POINTER(C.c_int)]),
    "pf_unet_pack_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64_p, C.c_int, C.c_void_p]),
    "pf_unet_pack_missing": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "pf_unet_bind_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "pf_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_unet_forward_prepared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(UNetPrepared),
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_unet_forward_cfg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(UNetPrepared),
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_unet_time_bias_width": (C.c_int, [C.c_void_p]),
    "pf_unet_cross_bias_width": (C.c_int, [C.c_void_p]),
    "pf_unet_prepare_time": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_unet_prepare_cond": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_unet_n_launches_prepared": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pf_unet_workspace_bytes_cfg": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "pf_unet_n_launches_cfg": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pf_unet_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pf_unet_get_option": (C.c_int, [C.c_void_p, C.c_int]),
    "pf_unet_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "pf_unet_track_absmax": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_unet_profile_read_direct": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "pf_unet_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), c_float_p, C.POINTER(C.c_double), C.c_int]),
    "pf_unet_n_launches": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pf_cfg_combine": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ddpm_step": (C.c_int, [C.c_void_p] * 6 + [C.POINTER(DdpmCoef), C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_axpby": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ddim_step": (C.c_int, [C.c_void_p] * 6 + [C.POINTER(DdimCoef), C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_randn": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    "pf_ddpm_step_rng": (C.c_int, [C.c_void_p] * 4 + [C.POINTER(DdpmCoef), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ddim_step_rng": (C.c_int, [C.c_void_p] * 5 + [C.POINTER(DdimCoef), C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ddpm_step_rng_dev": (C.c_int, [C.c_void_p] * 6 + [C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ddim_step_rng_dev": (C.c_int, [C.c_void_p] * 7 + [C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_clock_probe": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_mfma_probe": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "pf_step_state_set": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]),
    "pf_step_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_step_end": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "pf_randn_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p]),
    "pf_ddpm_step_dev": (C.c_int, [C.c_void_p] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ddim_step_dev": (C.c_int, [C.c_void_p] * 6 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pf_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pf_comm_bcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "pf_comm_destroy": (C.c_int, [C.c_void_p]),
    "pf_prmat2c_durations": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "pf_encoder_create": (C.c_int, [C.c_int] * 6 + [C.POINTER(C.c_void_p)]),
    "pf_encoder_destroy": (None, [C.c_void_p]),
    "pf_encoder_weight_bytes": (C.c_size_t, [C.c_void_p]),
    "pf_encoder_pack_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, c_i64_p, C.c_int, C.c_void_p]),
    "pf_encoder_pack_missing": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "pf_encoder_bind_weights": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_encoder_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "pf_encoder_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_packed_gemm_weight_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pf_pack_gemm_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pf_pack_gemm_weight_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pf_pack_upfold_weight_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pf_wino_weight_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "pf_pack_wino_weight_bf16x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pf_unet_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "pf_unet_get_precision": (C.c_int, [C.c_void_p]),
    "pf_x3_element": (C.c_int, []),
    "pf_gn_scale_shift": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_ln_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_ln_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_mlp_geglu_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_mlp_geglu_proj_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 10),
    "pf_conv2d": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "pf_conv_stats_tiles": (C.c_int, [C.POINTER(ConvArgs)]),
    "pf_conv_splitk_ws_bytes": (C.c_size_t, [C.POINTER(ConvArgs)]),
    "pf_gn_finalize_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_attention_bf16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pf_attention_split_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "pf_attention_bf16x3_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pf_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

_libs: dict = {}


def load(variant: Optional[str] = None) -> C.CDLL:
    """dlopen the library (default build, or the named variant) and attach prototypes.  Raises if it has not been built.
    Both builds can live in one process: they are opened RTLD_LOCAL and share nothing but the HIP runtime."""
    v = X3_VARIANT if variant is None else variant
    if v in _libs:
        return _libs[v]
    path = lib_path(v)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -m polyffusion_amd.build{' --variant=' + v if v else ''}` "
            "(or __graft_entry__.build()). There is no CPU fallback for this path.")
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pf_x3_element() != (1 if v == "f16" else 0):
        raise RuntimeError(f"{path}: built with the wrong split element type (pf_x3_element() = {lib.pf_x3_element()})")
    _libs[v] = lib
    return lib


def check(rc: int, what: str = "", lib: Optional[C.CDLL] = None) -> int:
    """Raise on a negative return code, with the message of the library that produced it (`lib`: default build if omitted)."""
    if rc < 0:
        msg = (lib or load()).pf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libpfhip {what}: error {rc}: {msg}")
    return rc


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("polyffusion_amd needs an AMD GPU (gfx950); no CPU fallback exists for the denoising path")


def ptr(t) -> int:
    """Raw device/host address of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
