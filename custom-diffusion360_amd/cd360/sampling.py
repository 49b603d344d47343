"""Native equivalent of what reference sample.py does to the UNet at inference time (sample.py:33-136,247-281).

sample.py monkey-patches SpatialTransformer.forward / BasicTransformerBlock.forward so that (1) the reference features come
from the per-block `references` buffer of the delta checkpoint, picked by the global `choices`, with `references[-1]` (the
null image) for the unconditional CFG third, and (2) the FeatureNeRF render runs once per image and is cached in
`rendered_feat`.  Those patched forwards also work on this package's modules (same attribute names).  This module offers
the same behaviour without patching, plus per-image residency the reference does not have: the cross-attention K / V of
the (constant) text context are projected once per image instead of once per block per step.
"""
from __future__ import annotations

from typing import Iterable, List

import torch


def pose_blocks(model: torch.nn.Module):
    """(name, block) for every transformer block that carries the pose path (has `pose_emb_layers`)."""
    return [(n, m) for n, m in model.named_modules() if hasattr(m, "pose_emb_layers") and hasattr(m, "reference_attn")]


def _cross_attentions(model: torch.nn.Module):
    return [m for m in model.modules() if hasattr(m, "cache_context_kv")]


def set_references(model: torch.nn.Module, references: dict) -> None:
    """Register `references` buffers ([N_train+1, hw, C], last row = null image) as sgm/util.py:231-235 does."""
    for name, blk in pose_blocks(model):
        ref = references[name]
        if "references" in blk._buffers:
            blk._buffers["references"] = ref
        else:
            blk.register_buffer("references", ref)


def enable_reference_sampling(model: torch.nn.Module, choices: Iterable[int], cache_context: bool = True) -> List[str]:
    """Switch every pose block to sample.py semantics with the given reference-view `choices` (sample.py:274-278).

    cache_context=True additionally keeps the cross-attention K / V projections of the text context resident between steps.
    Contract: the `context` buffer handed to the UNet is not rewritten in place-without-version-bump or re-allocated until
    `clear_rendered_feat(model)` is called (sample.py builds c/uc once per image and calls clear_rendered_feat between images)."""
    choices = [int(c) for c in choices]
    names = []
    for name, blk in pose_blocks(model):
        if not hasattr(blk, "references"):
            raise RuntimeError(f"{name} has no `references` buffer (load a delta checkpoint or call set_references)")
        blk.reference_choices = choices
        blk.rendered_feat = None
        names.append(name)
    for att in _cross_attentions(model):
        att.cache_context_kv = bool(cache_context)
        att._kv_cache = None
        att._kv8_cache = None
    return names


def disable_reference_sampling(model: torch.nn.Module) -> None:
    for _, blk in pose_blocks(model):
        blk.reference_choices = None
        blk.rendered_feat = None
    for att in _cross_attentions(model):
        att.cache_context_kv = False
        att._kv_cache = None
        att._kv8_cache = None


def clear_rendered_feat(model: torch.nn.Module) -> None:
    """DiffusionEngine.clear_rendered_feat (sgm/models/diffusion.py:165-170): call between images.  Also drops the per-image
    context K / V cache."""
    for _, blk in pose_blocks(model):
        blk.rendered_feat = None
        if hasattr(blk, "_rendered_proj"):
            blk._rendered_proj = None  # rendered_feat @ Wb^T kept beside the cached render
    for att in _cross_attentions(model):
        att._kv_cache = None
        att._kv8_cache = None
