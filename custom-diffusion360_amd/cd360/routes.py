"""Host-side route switches: which of two implementations of the SAME module code runs (every same-box A/B quoted in DESIGN.md section 7c
was made with them).  Each switch is resolved ONCE, when this module is imported, from its historical CD360_* environment variable -- the
hot paths read a plain attribute, never os.environ -- and harnesses flip them explicitly:

    from cd360 import routes
    with routes.override(library_linear=True): ...     # bench.py's train_step.library_ms, tests
    routes.set(no_cfg_dedup=True)

None is needed for normal use; every default (False) is the measured best."""
from __future__ import annotations

import os
from contextlib import contextmanager

# attribute -> environment variable read at import
_ENV = {
    "library_linear": "CD360_LIBRARY_LINEAR",          # F.linear (hipBLASLt) + add_layernorm / geglu kernels instead of cd360_gemm_bf16
    "no_a2_fuse": "CD360_NO_A2_FUSE",                  # text cross-attention as q GEMM + attn_smallk
    "no_qproj_attn": "CD360_NO_QPROJ_ATTN",            # pose-token cross-attention likewise
    "no_upsample_fold": "CD360_NO_UPSAMPLE_FOLD",      # Upsample as F.interpolate + conv3x3
    "no_render_commute": "CD360_NO_RENDER_COMMUTE",    # out projection per sample before the volume render
    "no_cfg_dedup": "CD360_NO_CFG_DEDUP",              # all three CFG thirds rendered
    "no_pose_proj_cache": "CD360_NO_POSE_PROJ_CACHE",  # rendered Wb^T recomputed every step
    "no_gn_stats": "CD360_NO_GN_STATS",                # GroupNorm statistics always by their own pass
    "no_concat_stats": "CD360_NO_CONCAT_STATS",        # ... only after a skip concat
    "no_emb_merge": "CD360_NO_EMB_MERGE",              # one GEMM per time-embedding projection
    "edge_convs_miopen": "CD360_EDGE_CONVS_MIOPEN",    # the 4 -> 320 and 320 -> 4 convolutions on MIOpen
    "no_train_fusions": "CD360_NO_TRAIN_FUSIONS",      # fine-tuning: derived FeatureNeRF weights, render loss terms and residual hand-offs op by op in torch
    "strict_sample_py": "CD360_STRICT_SAMPLE_PY",      # sample.py's rebound forwards are left in place and run themselves (the strict module route)
    "no_stage": "CD360_NO_STAGE",                      # the sampling job's captured steps with the denoiser / embedding scalar math as torch launches (round 5)
    "no_out_conv4": "CD360_NO_OUT_CONV4",              # the 320 -> 4 output convolution on the GEMM core with Cout padded to 16 (round 5) instead of cd360_out_conv4_bf16
    "no_host_glue": "CD360_NO_HOST_GLUE",              # fine-tuning Linears through the Python autograd node (grad.LinearFn) instead of the C++ one
    "fp8_attn": "CD360_FP8_ATTN",                      # BASELINE configs[4]: attention contractions of the fused cross-attention on fp8 MFMA
}

for _name, _var in _ENV.items():
    globals()[_name] = bool(os.environ.get(_var))


def set(**switches) -> None:
    for k, v in switches.items():
        if k not in _ENV:
            raise KeyError(f"unknown route switch {k}")
        globals()[k] = bool(v)


def get() -> dict:
    return {k: globals()[k] for k in _ENV}


@contextmanager
def override(**switches):
    saved = {k: globals()[k] for k in switches}
    set(**switches)
    try:
        yield
    finally:
        set(**saved)
