"""Reference dotted path `sgm.modules.diffusionmodules.denoiser` -> cd360.sampler."""
from cd360.sampler import DiscreteDenoiser  # noqa: F401
