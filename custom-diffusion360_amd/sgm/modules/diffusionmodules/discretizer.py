"""Reference dotted path `sgm.modules.diffusionmodules.discretizer` -> cd360.sampler."""
from cd360.sampler import LegacyDDPMDiscretization  # noqa: F401
