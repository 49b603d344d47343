"""Harness that imports the *reference's own* hot-path modules in this container.

Only used by tests/golden/make_golden.py (run once, here, where /root/reference exists) to
capture golden input/output vectors.  Nothing under tests/ that runs on the GPU box imports
this file, and no reference source or bytecode is ever copied into the repo.

Recipe: the reference cannot be imported as-is because its packages pull
pytorch_lightning / kornia / open_clip / pytorch3d / xformers / omegaconf, none of which
are installed.  So:
  1. register empty package objects `sgm`, `sgm.modules`, `sgm.modules.diffusionmodules`
     whose __path__ points into /root/reference, which skips the heavy __init__.py files;
  2. provide stand-ins for the third-party libraries:
       pytorch3d  -> cd360.cameras (the conventions restated from SURVEY.md Appendix B;
                     "parity unpinned" at this boundary),
       xformers.ops.memory_efficient_attention -> plain softmax attention in fp32,
       omegaconf.ListConfig -> a list subclass;
  3. remap device="cuda" in torch.linspace to CPU (Raymarcher.__init__ hard-codes it,
     nerfsd_pytorch3d.py:249,251).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("CD360_REFERENCE", "/root/reference")
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_REPO, "custom-diffusion360_amd"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _stub_third_party():
    from cd360 import cameras as cams

    # ---- pytorch3d ----
    p3d = types.ModuleType("pytorch3d")
    p3d.__path__ = []
    rend = types.ModuleType("pytorch3d.renderer")
    rend.__path__ = []
    cam_m = types.ModuleType("pytorch3d.renderer.cameras")
    cam_m.PerspectiveCameras = cams.PerspectiveCameras
    cu = types.ModuleType("pytorch3d.renderer.camera_utils")
    cu.join_cameras_as_batch = cams.join_cameras_as_batch
    impl = types.ModuleType("pytorch3d.renderer.implicit")
    impl.__path__ = []
    rs = types.ModuleType("pytorch3d.renderer.implicit.raysampling")

    class RayBundle:
        def __init__(self, origins, directions, lengths, xys):
            self.origins, self.directions, self.lengths, self.xys = origins, directions, lengths, xys

    def ray_bundle_to_ray_points(rb):
        return rb.origins[..., None, :] + rb.lengths[..., :, None] * rb.directions[..., None, :]

    rs.RayBundle = RayBundle
    rend.ray_bundle_to_ray_points = ray_bundle_to_ray_points
    rend.cameras, rend.camera_utils, rend.implicit = cam_m, cu, impl
    impl.raysampling = rs
    c_mod = types.ModuleType("pytorch3d._C")

    def sample_pdf(*a, **k):  # dead path
        raise NotImplementedError("importance sampling is dead code in the shipped config")

    c_mod.sample_pdf = sample_pdf
    p3d.renderer, p3d._C = rend, c_mod
    for name, m in [
        ("pytorch3d", p3d),
        ("pytorch3d.renderer", rend),
        ("pytorch3d.renderer.cameras", cam_m),
        ("pytorch3d.renderer.camera_utils", cu),
        ("pytorch3d.renderer.implicit", impl),
        ("pytorch3d.renderer.implicit.raysampling", rs),
        ("pytorch3d._C", c_mod),
    ]:
        sys.modules[name] = m

    # ---- xformers ----
    xf = types.ModuleType("xformers")
    xf.__path__ = []
    xops = types.ModuleType("xformers.ops")

    def memory_efficient_attention(q, k, v, attn_bias=None, op=None):
        assert attn_bias is None
        s = torch.einsum("bqd,bkd->bqk", q.float(), k.float()) * (q.shape[-1] ** -0.5)
        return torch.einsum("bqk,bkd->bqd", s.softmax(-1), v.float()).to(q.dtype)

    xops.memory_efficient_attention = memory_efficient_attention
    xf.ops = xops
    sys.modules["xformers"], sys.modules["xformers.ops"] = xf, xops

    # ---- omegaconf ----
    oc = types.ModuleType("omegaconf")
    oc.__path__ = []

    class ListConfig(list):
        pass

    oc.ListConfig = ListConfig
    oc.OmegaConf = type("OmegaConf", (), {})
    lc = types.ModuleType("omegaconf.listconfig")
    lc.ListConfig = ListConfig
    oc.listconfig = lc
    sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = oc, lc


_done = False


def import_reference():
    """Returns a namespace with the reference hot-path modules."""
    global _done
    sys.dont_write_bytecode = True
    if not _done:
        if not os.path.isdir(REF_ROOT):
            raise RuntimeError(f"{REF_ROOT} not present: golden vectors can only be generated in the build container")
        _stub_third_party()
        _pkg("sgm", os.path.join(REF_ROOT, "sgm"))
        _pkg("sgm.modules", os.path.join(REF_ROOT, "sgm", "modules"))
        _pkg("sgm.modules.diffusionmodules", os.path.join(REF_ROOT, "sgm", "modules", "diffusionmodules"))
        real_linspace = torch.linspace

        def linspace(*a, **k):
            if str(k.get("device", "")) == "cuda" and not torch.cuda.is_available():
                k["device"] = "cpu"
            return real_linspace(*a, **k)

        torch.linspace = linspace
        _done = True
    ns = types.SimpleNamespace()
    ns.util = importlib.import_module("sgm.util")
    ns.dutil = importlib.import_module("sgm.modules.diffusionmodules.util")
    ns.cameraray = importlib.import_module("sgm.modules.utils_cameraray")
    ns.nerf = importlib.import_module("sgm.modules.nerfsd_pytorch3d")
    ns.attention = importlib.import_module("sgm.modules.attention")
    ns.openaimodel = importlib.import_module("sgm.modules.diffusionmodules.openaimodel")
    ns.sampling = importlib.import_module("sgm.modules.diffusionmodules.sampling")
    ns.guiders = importlib.import_module("sgm.modules.diffusionmodules.guiders")
    ns.denoiser = importlib.import_module("sgm.modules.diffusionmodules.denoiser")
    ns.discretizer = importlib.import_module("sgm.modules.diffusionmodules.discretizer")
    return ns


def import_reference_loss():
    """sgm.modules.diffusionmodules.{loss, sigma_sampling} of the reference.  loss.py:6-7 imports LPIPS (torchvision) and
    GeneralConditioner (open_clip / kornia), neither installed nor on the l2 path: both names are stubbed."""
    ns = import_reference()
    for name, attr in (("sgm.modules.autoencoding.lpips.loss.lpips", "LPIPS"), ("sgm.modules.encoders.modules", "GeneralConditioner")):
        parts = name.split(".")
        for i in range(3, len(parts) + 1):
            sub = ".".join(parts[:i])
            if sub not in sys.modules:
                m = types.ModuleType(sub)
                m.__path__ = []
                sys.modules[sub] = m
        setattr(sys.modules[name], attr, type(attr, (torch.nn.Module,), {}))
    ns.loss = importlib.import_module("sgm.modules.diffusionmodules.loss")
    ns.sigma_sampling = importlib.import_module("sgm.modules.diffusionmodules.sigma_sampling")
    return ns
