"""Plugin surface kept from the reference (sgm/util.py:168-199): objects are built from
`{"target": "dotted.path.Class", "params": {...}}` nodes; works with plain dicts or OmegaConf nodes."""
from __future__ import annotations

import importlib
from inspect import isfunction

import torch


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def get_obj_from_str(string: str, reload: bool = False, invalidate_cache: bool = True):
    module, cls = string.rsplit(".", 1)
    if invalidate_cache:
        importlib.invalidate_caches()
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x, target_dims):
    dims_to_append = target_dims - x.ndim
    if dims_to_append < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * dims_to_append]


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total


def load_yaml_config(path: str):
    """OmegaConf-free loader for the reference's YAML configs (omegaconf is optional at run time)."""
    try:
        from omegaconf import OmegaConf

        return OmegaConf.load(path)
    except ImportError:
        import yaml

        with open(path) as f:
            return yaml.safe_load(f)
