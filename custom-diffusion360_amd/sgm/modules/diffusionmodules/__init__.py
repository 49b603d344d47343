"""The reference's dotted paths for the sampler stack, resolved without one re-export file per name.

The YAML of the reference names its sampler / guider / denoiser classes by module path (`configs/train_co3d_concept.yaml:14-25,119-131`:
`sgm.modules.diffusionmodules.denoiser.DiscreteDenoiser`, `...guiders.ScheduledCFGImgTextRef`, `...sampling.EulerEDMSampler`, ...), and
`instantiate_from_config` imports that path (`sgm/util.py:168-185`).  All of those classes live in ONE module here, `cd360.sampler`; the
six reference module names are registered as aliases of it, so `importlib.import_module("sgm.modules.diffusionmodules.guiders")` and
`from sgm.modules.diffusionmodules.sampling import EulerEDMSampler` both work."""
import sys as _sys

from cd360 import sampler as _sampler

for _name in ("denoiser", "denoiser_scaling", "denoiser_weighting", "discretizer", "guiders", "sampling"):
    _sys.modules[f"{__name__}.{_name}"] = _sampler
    globals()[_name] = _sampler
