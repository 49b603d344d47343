"""Training-time noise-level samplers (reference sgm/modules/diffusionmodules/sigma_sampling.py:6-54), YAML targets of
`loss_fn_config` (configs/train_co3d_concept.yaml:119-134).  Host-side, a handful of scalars per step."""
import torch

from ...util import default, instantiate_from_config


class EDMSampling:
    """log-normal sigma: exp(p_mean + p_std * N(0,1)) (:6-13)."""

    def __init__(self, p_mean=-1.2, p_std=1.2):
        self.p_mean, self.p_std = p_mean, p_std

    def __call__(self, n_samples, rand=None):
        return (self.p_mean + self.p_std * default(rand, torch.randn((n_samples,)))).exp()


class _TableSampling:
    def __init__(self, discretization_config, num_idx, do_append_zero=False, flip=True):
        self.num_idx = num_idx
        self.sigmas = instantiate_from_config(discretization_config)(num_idx, do_append_zero=do_append_zero, flip=flip)

    def idx_to_sigma(self, idx):
        return self.sigmas[idx]


class DiscreteSampling(_TableSampling):
    """uniform index into the discretisation's sigma table (:16-32); used for the reference views with num_idx=50."""

    def __init__(self, discretization_config, num_idx, num_idx_start=0, do_append_zero=False, flip=True):
        super().__init__(discretization_config, num_idx, do_append_zero, flip)
        self.num_idx_start = num_idx_start

    def __call__(self, n_samples, rand=None):
        return self.idx_to_sigma(default(rand, torch.randint(self.num_idx_start, self.num_idx, (n_samples,))))


class CubicSampling(_TableSampling):
    """index = floor((1 - u^3) (num_idx - 1)), u ~ U(0,1): biased to the noisy end (:35-54).  As in the reference the uniform
    draw is made even when `rand` overrides the index, so the RNG stream stays aligned."""

    def __call__(self, n_samples, rand=None):
        t = torch.rand((n_samples,))
        t = ((1 - t**3) * (self.num_idx - 1)).long()
        return self.idx_to_sigma(default(rand, t))
