"""Reference dotted path `sgm.modules.diffusionmodules.guiders` -> cd360.sampler."""
from cd360.sampler import IdentityGuider, ScheduledCFGImgTextRef, VanillaCFGImgRef  # noqa: F401
