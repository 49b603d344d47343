"""Drop-in `sgm` package for the custom-diffusion360 hot path (MI355X / HIP).

Only the modules on the pose-conditioned denoising path exist here: put this directory ahead of the
reference's on sys.path and the YAML `target:` strings of configs/train_co3d_concept.yaml:27-54 resolve to these
classes.  Data loading, text encoders, VAE, Lightning engine and samplers stay the reference's own (out of scope)."""
from .util import instantiate_from_config  # noqa: F401
