"""Deterministic synthetic weights/inputs shared by the golden generator and the tests.

Weights are never stored in fixtures: both sides regenerate them from the tensor *name* and
shape with numpy's PCG64 (stable across numpy versions for `standard_normal`), so a fixture
holds only inputs that are not regenerable and the reference's outputs.
Random (non-zero) values are used for `pose_emb_layers`, `decoder` and `proj_out`, which the
reference zero/identity-initialises (attention.py:515-516,795; nerfsd_pytorch3d.py:49-51) --
with the stock init the pose path is a no-op and parity would be vacuous.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def tensor(name: str, shape, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    return torch.from_numpy((_rng(name, seed).standard_normal(tuple(shape)) * scale).astype(np.float32))


def uniform(name: str, shape, seed: int = 0) -> torch.Tensor:
    return torch.from_numpy(_rng(name, seed).random(tuple(shape)).astype(np.float32))


def synth_state_dict(shapes: dict, seed: int = 0) -> dict:
    """name -> shape  =>  name -> fp32 tensor."""
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        leaf = name.split(".")[-1]
        is_norm = ("norm" in name.split(".")[-2]) or name.endswith(("in_layers.0.weight", "in_layers.0.bias", "out_layers.0.weight",
                                                                     "out_layers.0.bias", "out.0.weight", "out.0.bias"))
        if "raymarcher" in name:
            continue  # buffers are recomputed, not weights
        if leaf == "bias" or len(shape) == 1:
            if is_norm and leaf == "weight":
                out[name] = 1.0 + tensor(name, shape, seed, 0.1)
            else:
                out[name] = tensor(name, shape, seed, 0.05)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = tensor(name, shape, seed, 1.0 / np.sqrt(fan_in))
        if name.endswith("pose_emb_layers.weight"):
            c = shape[0]
            out[name] = torch.cat([torch.eye(c), torch.zeros(c, c)], 1) + tensor(name, shape, seed, 0.05)
        if name.endswith("decoder.weight"):
            out[name] = tensor(name, shape, seed, 0.3 / np.sqrt(shape[1]))
    return out


def load_into(module: torch.nn.Module, seed: int = 0) -> dict:
    """Fill `module` with synth weights (by its own state_dict names); returns the dict used."""
    sd = module.state_dict()
    new = synth_state_dict({k: v.shape for k, v in sd.items()}, seed)
    missing = module.load_state_dict(new, strict=False)
    assert not missing.unexpected_keys
    assert all("raymarcher" in k for k in missing.missing_keys), missing.missing_keys
    return new
