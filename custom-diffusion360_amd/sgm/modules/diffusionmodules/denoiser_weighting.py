"""Reference dotted path `sgm.modules.diffusionmodules.denoiser_weighting` -> cd360.sampler."""
from cd360.sampler import EpsWeighting  # noqa: F401
