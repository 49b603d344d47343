"""Reference dotted path `sgm.modules.diffusionmodules.denoiser_scaling` -> cd360.sampler."""
from cd360.sampler import EpsScaling  # noqa: F401
