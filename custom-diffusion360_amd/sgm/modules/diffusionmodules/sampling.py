"""Reference dotted path `sgm.modules.diffusionmodules.sampling` -> cd360.sampler (see that module)."""
from cd360.sampler import EulerEDMSampler  # noqa: F401
