"""ctypes binding of libcd360_hip.so (the C ABI declared in include/cd360_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises.
The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950)."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CD360_LIB: another build of the SAME library (tools/probe/whatif_build.sh, gemm_stamp.sh) -- a path, never a fallback
LIB_PATH = os.environ.get("CD360_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libcd360_hip.so")

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
    "cd360_sample_pdf": (c_int, [_P, _P, _P, _P, _P, c_float, c_int64, c_int, c_int, _P]),
    "cd360_feature_gather": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_plucker_features": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_plucker_features_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_nerf_k_padded": (c_int, []),
    "cd360_nerf_mlp_aggregate": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_nerf_ws_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "cd360_nerf_mlp_aggregate_ws": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    "cd360_nerf_mlp_aggregate_bwd": (c_int, [_P, _P, _P, _P, c_int] + [_P] * 15 + [c_int] * 5 + [_P]),
    "cd360_nerf_mlp_aggregate_bwd_det": (c_int, [_P, _P, _P, _P, c_int] + [_P] * 16 + [c_int] * 5 + [_P]),
    "cd360_volrender": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_volrender_bwd": (c_int, [_P, _P, _P, _P, c_int] + [_P] * 8 + [c_int] * 6 + [_P]),
    "cd360_rowdot4_bf16": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cd360_rowdot1_bf16": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cd360_render_loss_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_render_loss_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "cd360_nerf_pack_weights_bf16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "cd360_nerf_unpack_grads_bf16": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int), _P, _P, _P, c_int, c_int, _P]),
    "cd360_geglu_bf16": (c_int, [_P, _P, c_int64, c_int, _P]),
    "cd360_concat_channels_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    "cd360_add_layernorm_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "cd360_geglu_bwd_bf16": (c_int, [_P, _P, _P, c_int64, c_int, _P]),
    "cd360_add_layernorm_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "cd360_gn_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "cd360_gn_silu_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "cd360_cfg_euler_step_f32": (c_int, [_P, _P, _P, _P, c_float, c_float, _P, c_int64, _P]),
    "cd360_unet_stage_in": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_cfg_euler_step_cl": (c_int, [_P, _P, _P, _P, c_float, c_float, c_int, c_int64, c_int, _P]),
    "cd360_out_conv4_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "cd360_conv_k_order": (c_int, [c_int, c_int]),
    "cd360_conv_igemm_bf16": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "cd360_conv_stats_slabs": (c_int, [c_int]),
    "cd360_conv_stats_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "cd360_conv_dma_slab_rows": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "cd360_conv3x3_dma_bf16": (c_int, [_P, _P, _P, _P, c_int64, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "cd360_conv_up2x_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cd360_pose_embed_bf16": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "cd360_gemm_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, c_int, c_int, c_float, _P, _P, c_int, _P]),
    "cd360_qproj_attn_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, c_float, _P, _P, _P,
                                      c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_float, _P]),
    "cd360_qproj_attn_dedup_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, c_float, _P, _P, _P,
                                            c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_float, c_int, _P]),
    "cd360_kv_fp8_bytes": (c_int64, [c_int, c_int]),
    "cd360_kv_pack_fp8": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64, _P]),
    "cd360_qproj_attn_fp8_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, c_float, _P, _P, _P,
                                          c_int, c_int, c_float, c_int, _P]),
    "cd360_gemm_tile_n": (c_int, [c_int64, c_int]),
    "cd360_gemm_cstats_rows": (c_int, [c_int64, c_int]),
    "cd360_gemm_cstats_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P]),
    "cd360_row_stats_bf16": (c_int, [_P, _P, c_int64, c_int, c_int64, _P]),
    "cd360_gn_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "cd360_gn_silu_bf16": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_int, _P, c_int, _P]),
    "cd360_rowdot4_bwd_slabs": (c_int, [c_int64]),
    "cd360_rowdot4_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "cd360_gemm_tn_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "cd360_gemm_tn_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int64, c_int64, c_int, _P, _P]),
    "cd360_adamw_tick": (c_int, [_P, _P]),
    "cd360_adamw_bf16": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, _P]),
    "cd360_prefetch_arm": (c_int, [_P, c_int, c_int, c_int64, _P]),
    "cd360_prefetch_disarm": (c_int, []),
    "cd360_prefetch_arm_on": (c_int, [_P, _P, c_int, c_int, c_int64, _P]),
    "cd360_prefetch_disarm_on": (c_int, [_P]),
    "cd360_set_stream_tuning": (c_int, [_P, _P]),
    "cd360_get_stream_tuning": (c_int, [_P, _P]),
    "cd360_query_stream": (c_int, [_P]),
    "cd360_set_tuning": (c_int, [_P]),
    "cd360_get_tuning": (c_int, [_P]),
    "cd360_whatif_build": (c_int, []),
}

TUNING_FIELDS = ("gemm_cfg", "gemm_group_m", "gemm_movers", "gemm_ksplit", "conv_cfg", "conv_dma", "conv_kgroup", "conv_wide", "conv_wmajor",
                 "conv_split", "attn_smallk", "attn_smallk_wgs", "attn_self", "attn_fast", "nerf_kernel", "qattn_cfg", "whatif", "gemm_small", "qattn_keys16",
                 "qattn_split", "store_wt", "conv_halo", "gemm_asm4")


class Tuning(ctypes.Structure):
    """struct cd360_tuning of include/cd360_hip.h: -1 = choose by shape (the default of every field)."""
    _fields_ = [("size", ctypes.c_int32)] + [(f, ctypes.c_int32) for f in TUNING_FIELDS] + [("reserved", ctypes.c_int32 * 2)]


# environment variable -> tuning field: read ONCE, when the library is loaded (the C side never reads the environment)
TUNING_ENV = {
    "CD360_GEMM_CFG": "gemm_cfg", "CD360_GEMM_GROUP_M": "gemm_group_m", "CD360_GEMM_MOVERS": "gemm_movers", "CD360_GEMM_KSPLIT": "gemm_ksplit",
    "CD360_CONV_CFG": "conv_cfg", "CD360_CONV_DMA": "conv_dma", "CD360_CONV_KGROUP": "conv_kgroup", "CD360_CONV_WIDE": "conv_wide",
    "CD360_CONV_WMAJOR": "conv_wmajor", "CD360_CONV_SPLIT": "conv_split", "CD360_ATTN_SMALLK": "attn_smallk", "CD360_SMALLK_WGS": "attn_smallk_wgs",
    "CD360_ATTN_SELF": "attn_self", "CD360_ATTN_FAST": "attn_fast", "CD360_NERF_KERNEL": "nerf_kernel", "CD360_CONV_HALO": "conv_halo", "CD360_QATTN_CFG": "qattn_cfg",
    "CD360_GEMM_ABL": "whatif", "CD360_GEMM_SMALL": "gemm_small", "CD360_QATTN_KEYS16": "qattn_keys16", "CD360_QATTN_SPLIT": "qattn_split",
    "CD360_STORE_WT": "store_wt", "CD360_GEMM_ASM4": "gemm_asm4",
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
    # torch FIRST: its wheel carries its own HIP runtime (torch/lib/libamdhip64.so), and the device pointers and streams this library
    # is handed belong to THAT runtime.  Loaded before torch, libcd360_hip.so would pull in the system runtime (/opt/rocm/lib) as a second
    # copy in the process and every launch on torch's streams would fail ("HIP launch failure": build() followed by smoke() in one
    # process did exactly that).  With torch's runtime already mapped, the dynamic loader binds this library's libamdhip64 dependency to it.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    if os.environ.get("CD360_LIB"):
        check_symbols = False  # an explicitly named older / probe build (same-box A/B): entry points it predates stay unbound
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            if check_symbols:
                raise Cd360Error(f"libcd360_hip.so does not export {name}") from e
            continue
        fn.restype, fn.argtypes = res, args
    _lib = lib
    env = {}
    for name, field in TUNING_ENV.items():
        raw = os.environ.get(name, "")
        if raw != "":
            try:
                env[field] = int(raw)
            except ValueError:
                raise Cd360Error(f"{name}={raw!r}: tuning variables are integers (include/cd360_hip.h: cd360_tuning.{field}); read once, when "
                                 "the library loads -- later changes of os.environ have no effect, use cd360._lib.set_tuning()") from None
    if env:
        set_tuning(**env)
    return lib


def get_tuning() -> dict:
    t = Tuning()
    check(load().cd360_get_tuning(ctypes.byref(t)), "cd360_get_tuning")
    return {f: getattr(t, f) for f in TUNING_FIELDS}


def set_tuning(**fields) -> None:
    """Override tiling / kernel choices of the C ABI (include/cd360_hip.h: cd360_tuning); unnamed fields keep their value, -1 restores
    a field's default; set_tuning() with no arguments changes nothing, reset_tuning() restores every default."""
    lib = _lib if _lib is not None else load()
    t = Tuning()
    check(lib.cd360_get_tuning(ctypes.byref(t)), "cd360_get_tuning")
    for k, v in fields.items():
        if k not in TUNING_FIELDS:
            raise KeyError(f"unknown tuning field {k}")
        setattr(t, k, int(v))
    t.size = ctypes.sizeof(Tuning)
    check(lib.cd360_set_tuning(ctypes.byref(t)), "cd360_set_tuning")


# ---- per-stream tuning (cd360_set_stream_tuning): launches on ONE stream read their own struct; the default above stays what it is ----
_stream_overrides = set()  # raw stream handles with an override (so that the hot path can skip cd360_query_stream while there is none)


def _stream_handle(stream) -> int:
    return int(getattr(stream, "cuda_stream", stream))


def set_stream_tuning(stream, **fields) -> None:
    """Tuning override for launches issued on `stream` (a torch.cuda.Stream or a raw handle): starts from the stream's current effective
    tuning, sets the named fields.  Two samplers / captures on two streams can hold different tilings in one process."""
    lib = _lib if _lib is not None else load()
    h = _stream_handle(stream)
    t = Tuning()
    check(lib.cd360_get_stream_tuning(ctypes.c_void_p(h), ctypes.byref(t)), "cd360_get_stream_tuning")
    for k, v in fields.items():
        if k not in TUNING_FIELDS:
            raise KeyError(f"unknown tuning field {k}")
        setattr(t, k, int(v))
    t.size = ctypes.sizeof(Tuning)
    check(lib.cd360_set_stream_tuning(ctypes.c_void_p(h), ctypes.byref(t)), "cd360_set_stream_tuning")
    _stream_overrides.add(h)


def get_stream_tuning(stream) -> dict:
    t = Tuning()
    check(load().cd360_get_stream_tuning(ctypes.c_void_p(_stream_handle(stream)), ctypes.byref(t)), "cd360_get_stream_tuning")
    return {f: getattr(t, f) for f in TUNING_FIELDS}


def clear_stream_tuning(stream) -> None:
    h = _stream_handle(stream)
    check(load().cd360_set_stream_tuning(ctypes.c_void_p(h), None), "cd360_set_stream_tuning")
    _stream_overrides.discard(h)
    if not _stream_overrides:
        load().cd360_query_stream(None)


def query_stream(handle) -> None:
    """Before a shape query (cd360_gemm_tile_n, ...): answer for launches on this stream.  Free while no stream has an override."""
    if _stream_overrides:
        _lib.cd360_query_stream(handle)


def reset_tuning() -> None:
    check(load().cd360_set_tuning(None), "cd360_set_tuning")


class tuning:
    """with tuning(gemm_cfg=4, gemm_movers=0): ...  -- the fields are restored on exit (A/B harnesses, tests)."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        self.saved = get_tuning()
        set_tuning(**self.fields)
        return self

    def __exit__(self, *a):
        set_tuning(**self.saved)
        return False


_ERR = {-1: "invalid argument (null/misaligned pointer or non-positive size)", -2: "unsupported shape or stride", -3: "HIP launch failure"}


def check(code: int, what: str) -> None:
    if code != 0:
        raise Cd360Error(f"{what} failed: {_ERR.get(code, code)}")


class TuningEnv:
    """Dict-like view of the tuning struct under the historical CD360_* variable names (tools/bench_gemm.py's A/B loops were written
    against os.environ): `ENV["CD360_GEMM_CFG"] = "4"` sets gemm_cfg = 4 through cd360_set_tuning, `ENV.pop(name, None)` restores
    the field's default.  Names that are not tuning fields go to os.environ unchanged."""

    def __setitem__(self, name, value):
        if name in TUNING_ENV:
            set_tuning(**{TUNING_ENV[name]: int(value)})
        else:
            os.environ[name] = value

    def pop(self, name, default=None):
        if name in TUNING_ENV:
            set_tuning(**{TUNING_ENV[name]: -1})
            return default
        return os.environ.pop(name, default)

    def get(self, name, default=None):
        if name in TUNING_ENV:
            v = get_tuning()[TUNING_ENV[name]]
            return default if v < 0 else str(v)
        return os.environ.get(name, default)


ENV = TuningEnv()
