"""Loader of the C++ host glue (csrc_host/cd360_host.cpp -> lib/_cd360_host.so): the autograd node of a Linear on the fine-tuning path as
C++ around the same C-ABI launches cd360/grad.py::LinearFn issues from Python.  It is glue above the ABI, not a compute path: it takes its
function pointers from the libcd360_hip.so that cd360._lib has loaded and refuses to initialise without it.  When the module has not been
built (or `cd360.routes.no_host_glue` is set: the A/B partner) the Python node runs -- same launches, more interpreter time."""
from __future__ import annotations

import importlib.util
import os

_mod = None
_tried = False


def get():
    """The initialised module, or None."""
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    from . import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "_cd360_host.so")
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (its libraries must be in the process before the extension resolves them)
    _lib.load()
    spec = importlib.util.spec_from_file_location("_cd360_host", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.init(_lib.LIB_PATH)
    _mod = mod
    return _mod
