"""Native equivalent of what reference sample.py does to the UNet at inference time (sample.py:33-136,247-281).

sample.py monkey-patches SpatialTransformer.forward / BasicTransformerBlock.forward so that (1) the reference features come
from the per-block `references` buffer of the delta checkpoint, picked by the global `choices`, with `references[-1]` (the
null image) for the unconditional CFG third, and (2) the FeatureNeRF render runs once per image and is cached in
`rendered_feat`.  Those patched forwards also work on this package's modules (same attribute names).  This module offers
the same behaviour without patching, plus one thing the reference cannot do: because the tables Y / lv of the fused
render depend only on (`references`, weights) they are computed once per model, not once per image.
"""
from __future__ import annotations

from typing import Iterable, List

import torch


def pose_blocks(model: torch.nn.Module):
    """(name, block) for every transformer block that carries the pose path (has `pose_emb_layers`)."""
    return [(n, m) for n, m in model.named_modules() if hasattr(m, "pose_emb_layers") and hasattr(m, "reference_attn")]


def set_references(model: torch.nn.Module, references: dict) -> None:
    """Register `references` buffers ([N_train+1, hw, C], last row = null image) as sgm/util.py:231-235 does."""
    for name, blk in pose_blocks(model):
        ref = references[name]
        if "references" in blk._buffers:
            blk._buffers["references"] = ref
        else:
            blk.register_buffer("references", ref)


def enable_reference_sampling(model: torch.nn.Module, choices: Iterable[int]) -> List[str]:
    """Switch every pose block to sample.py semantics with the given reference-view `choices` (sample.py:274-278)."""
    choices = [int(c) for c in choices]
    names = []
    for name, blk in pose_blocks(model):
        if not hasattr(blk, "references"):
            raise RuntimeError(f"{name} has no `references` buffer (load a delta checkpoint or call set_references)")
        blk.reference_choices = choices
        blk.rendered_feat = None
        names.append(name)
    return names


def disable_reference_sampling(model: torch.nn.Module) -> None:
    for _, blk in pose_blocks(model):
        blk.reference_choices = None
        blk.rendered_feat = None


def clear_rendered_feat(model: torch.nn.Module) -> None:
    """DiffusionEngine.clear_rendered_feat (sgm/models/diffusion.py:165-170): call between images."""
    for _, blk in pose_blocks(model):
        blk.rendered_feat = None
