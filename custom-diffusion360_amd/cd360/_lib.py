"""ctypes binding of libcd360_hip.so (the C ABI declared in include/cd360_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises.
The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950)."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libcd360_hip.so")

_P = c_void_p
_F32P = ctypes.POINTER(ctypes.c_float)
_I64P = ctypes.POINTER(c_int64)

# name -> (restype, argtypes); must list every symbol of include/cd360_hip.h
SIGNATURES = {
    "cd360_attn_fwd_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _I64P, _I64P, _I64P, _I64P, c_float, _P]),
    "cd360_attn_fwd_prescaled_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _I64P, _I64P, _I64P, _I64P, _P]),
    "cd360_attn_fwd_fp8mfma_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _I64P, _I64P, _I64P, _I64P, c_float, _F32P, _P]),
    "cd360_attn_fwd_lse_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _I64P, _I64P, _I64P, _I64P, c_float, _P]),
    "cd360_attn_bwd_bf16": (c_int, [_P] * 10 + [c_int] * 4 + [_I64P] * 8 + [c_float, _P]),
    "cd360_attn_fwd_xformers_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "cd360_patch_rays": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_ray_project_index": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "cd360_feature_gather": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_plucker_features": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_plucker_features_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_nerf_k_padded": (c_int, []),
    "cd360_nerf_mlp_aggregate": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_nerf_mlp_aggregate_bwd": (c_int, [_P, _P, _P, _P, c_int] + [_P] * 15 + [c_int] * 5 + [_P]),
    "cd360_volrender": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_volrender_bwd": (c_int, [_P, _P, _P, _P, c_int] + [_P] * 8 + [c_int] * 6 + [_P]),
    "cd360_rowdot4_bf16": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cd360_geglu_bf16": (c_int, [_P, _P, c_int64, c_int, _P]),
    "cd360_concat_channels_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    "cd360_add_layernorm_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "cd360_geglu_bwd_bf16": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cd360_add_layernorm_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "cd360_gn_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "cd360_gn_silu_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "cd360_cfg_euler_step_f32": (c_int, [_P, _P, _P, _P, c_float, c_float, _P, c_int64, _P]),
    "cd360_conv_k_order": (c_int, [c_int, c_int]),
    "cd360_conv_igemm_bf16": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "cd360_conv_stats_slabs": (c_int, [c_int]),
    "cd360_conv_stats_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "cd360_conv_dma_slab_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "cd360_conv3x3_dma_bf16": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "cd360_pose_embed_bf16": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "cd360_gemm_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, c_int, c_int, c_float, _P, _P, c_int, _P]),
    "cd360_qproj_attn_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, c_float, _P, _P, _P,
                                      c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_float, _P]),
    "cd360_qproj_attn_dedup_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, c_float, _P, _P, _P,
                                            c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_float, c_int, _P]),
    "cd360_gemm_tile_n": (c_int, [c_int64, c_int]),
    "cd360_gemm_cstats_rows": (c_int, [c_int64, c_int]),
    "cd360_gemm_cstats_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P]),
    "cd360_row_stats_bf16": (c_int, [_P, _P, c_int64, c_int, c_int64, _P]),
    "cd360_gn_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "cd360_gn_silu_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_int, _P]),
}

_lib = None


class Cd360Error(RuntimeError):
    pass


def load(check_symbols: bool = True):
    """dlopen the library and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Cd360Error(
            f"{LIB_PATH} not found: the HIP extension has not been built (run `python __graft_entry__.py`). "
            "There is no CPU or PyTorch fallback for the cd360 operators."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if check_symbols:
                raise Cd360Error(f"libcd360_hip.so does not export {name}") from e
            continue
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


_ERR = {-1: "invalid argument (null/misaligned pointer or non-positive size)", -2: "unsupported shape or stride", -3: "HIP launch failure"}


def check(code: int, what: str) -> None:
    if code != 0:
        raise Cd360Error(f"{what} failed: {_ERR.get(code, code)}")
