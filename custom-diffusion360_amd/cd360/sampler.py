"""Sampler / guider / denoiser step on either side of the UNet, restated from scratch.

Reference: EulerEDMSampler (sgm/modules/diffusionmodules/sampling.py:23-136,314-318), ScheduledCFGImgTextRef / VanillaCFGImgRef
(guiders.py:102-166), DiscreteDenoiser + EpsScaling (denoiser.py:6-79, denoiser_scaling.py:26-32), LegacyDDPMDiscretization
(discretizer.py:17-69).  Same class names and call signatures (they are re-exported under the reference's dotted paths in
custom-diffusion360_amd/sgm/modules/diffusionmodules/), so the YAML sampler/denoiser/guider configs resolve unchanged.

Everything stays on the device and nothing synchronises: the sigma -> index quantisation is an argmin + gather on the GPU, and
with `fused=True` the per-step tail (c_out scaling, 3-way CFG combine, to_d, Euler update) is one HIP kernel
(cd360_cfg_euler_step_f32) instead of ~10 tiny elementwise launches.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    return x[(...,) + (None,) * (target_dims - x.ndim)]


# ----------------------------------------------------------------------------------------------- discretisation
class LegacyDDPMDiscretization:
    """sigma_i = sqrt((1 - acp_i) / acp_i) on the SD linear-sqrt beta schedule (discretizer.py:47-69)."""

    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2).numpy()
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            timesteps = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]  # discretizer.py:11-14
            acp = self.alphas_cumprod[timesteps]
        elif n == self.num_timesteps:
            acp = self.alphas_cumprod
        else:
            raise ValueError
        sigmas = torch.tensor((1 - acp) / acp, dtype=torch.float32, device=device) ** 0.5
        return torch.flip(sigmas, (0,))

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        if do_append_zero:
            sigmas = torch.cat([sigmas, sigmas.new_zeros([1])])
        return sigmas if not flip else torch.flip(sigmas, (0,))


# ----------------------------------------------------------------------------------------------- denoiser
class EpsScaling:
    def __call__(self, sigma):
        return torch.ones_like(sigma), -sigma, 1 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()


class EpsWeighting:
    def __call__(self, sigma):
        return sigma ** -2.0


class DiscreteDenoiser(torch.nn.Module):
    """D(x; sigma) = c_skip x + c_out F(c_in x; idx(sigma)) with sigma snapped to the 1000-entry table (denoiser.py:22-79).
    The reference-stream noise injection of the training path (:26-39) is kept."""

    def __init__(self, weighting_config=None, scaling_config=None, num_idx=1000, discretization_config=None, do_append_zero=False,
                 quantize_c_noise=True, flip=True):
        super().__init__()
        from sgm.util import instantiate_from_config
        self.weighting = instantiate_from_config(weighting_config) if weighting_config else EpsWeighting()
        self.scaling = instantiate_from_config(scaling_config) if scaling_config else EpsScaling()
        disc = instantiate_from_config(discretization_config) if discretization_config else LegacyDDPMDiscretization()
        self.register_buffer("sigmas", disc(num_idx, do_append_zero=do_append_zero, flip=flip))
        self.quantize_c_noise = quantize_c_noise

    def w(self, sigma):
        return self.weighting(sigma)

    def sigma_to_idx(self, sigma):
        return (sigma - self.sigmas[:, None]).abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self.sigmas[idx]

    def possibly_quantize_sigma(self, sigma):
        return self.idx_to_sigma(self.sigma_to_idx(sigma))

    def possibly_quantize_c_noise(self, c_noise):
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise

    def network_inputs(self, input, sigma, kwargs):
        """Everything before the network call: (scaled input, c_noise, c_skip, c_out, kwargs)."""
        sigma = self.possibly_quantize_sigma(sigma)
        sigma_shape = sigma.shape
        sigma = append_dims(sigma, input.ndim)
        kwargs = dict(kwargs)
        sigmas_ref = kwargs.pop("sigmas_ref", None)
        if sigmas_ref is not None:
            kwargs["sigmas_ref"] = sigmas_ref
            if kwargs.get("input_ref") is not None:
                xr = kwargs["input_ref"]
                xr = xr + torch.randn_like(xr) * append_dims(sigmas_ref, xr.ndim)
                _, _, c_in_ref, _ = self.scaling(append_dims(sigmas_ref, xr.ndim))
                kwargs["input_ref"] = xr * c_in_ref
                kwargs["sigmas_ref"] = self.possibly_quantize_c_noise(sigmas_ref)
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(sigma_shape))
        return input * c_in, c_noise, c_skip, c_out, kwargs

    def forward(self, network, input, sigma, cond, sigmas_ref=None, **kwargs):
        if sigmas_ref is not None:
            kwargs["sigmas_ref"] = sigmas_ref
        x_in, c_noise, c_skip, c_out, kw = self.network_inputs(input, sigma, kwargs)
        predict, fg_mask_list, alphas_list, rgb_list = network(x_in, c_noise, cond, **kw)
        return predict * c_out + input * c_skip, fg_mask_list, alphas_list, rgb_list


# ----------------------------------------------------------------------------------------------- guiders
def _split_cat(c, uc, nx, order):
    """rows [0:nx] are the target's conditioning, the rest the reference views' (guiders.py:119-128)."""
    parts = {"uc1": uc[:nx], "uc2": uc[nx:], "c1": c[:nx], "c2": c[nx:]}
    return torch.cat([parts[k] for k in order], 0)


class ScheduledCFGImgTextRef:
    """3-way CFG: x_u + scale (x_c - x_ic) + scale_im (x_ic - x_u)   (guiders.py:102-133)."""

    branches = 3

    def __init__(self, scale: float, scale_im: float):
        self.scale, self.scale_im = scale, scale_im

    def __call__(self, x, sigma):
        x_u, x_ic, x_c = x.chunk(3)
        return x_u + self.scale * (x_c - x_ic) + self.scale_im * (x_ic - x_u)

    def prepare_inputs(self, x, s, c, uc):
        c_out = {}
        for k in c:
            if k in ("vector", "crossattn", "concat"):
                c_out[k] = _split_cat(c[k], uc[k], x.size(0), ("uc1", "uc1", "c1", "uc2", "c2", "c2"))
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 3), torch.cat([s] * 3), c_out


class VanillaCFGImgRef:
    """2-way CFG: x_u + scale (x_c - x_u)   (guiders.py:136-166)."""

    branches = 2

    def __init__(self, scale: float):
        self.scale = scale

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return x_u + self.scale * (x_c - x_u)

    def prepare_inputs(self, x, s, c, uc):
        c_out = {}
        for k in c:
            if k in ("vector", "crossattn", "concat"):
                c_out[k] = _split_cat(c[k], uc[k], x.size(0), ("uc1", "c1", "uc2", "c2"))
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


class IdentityGuider:
    branches = 1

    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, {k: c[k] for k in c}


# ----------------------------------------------------------------------------------------------- sampler
class EulerEDMSampler:
    """Euler steps over the sub-sampled sigma schedule, s_churn = 0 (DDIM-equivalent with EpsScaling)   (sampling.py:85-136,314-318).

    `denoiser(x_batched, sigma_batched, cond) -> (denoised, fg_masks, alphas, rgb_list)` is what DiffusionEngine.sample builds
    (sgm/models/diffusion.py:375-401)."""

    def __init__(self, discretization_config=None, num_steps: Optional[int] = None, guider_config=None, verbose: bool = False,
                 device: str = "cuda", s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
        from sgm.util import instantiate_from_config
        if s_churn != 0.0:
            raise NotImplementedError("s_churn > 0 (stochastic sampling) is not used by sample.py")
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config) if discretization_config else LegacyDDPMDiscretization()
        self.guider = instantiate_from_config(guider_config) if guider_config else IdentityGuider()
        self.verbose, self.device = verbose, device
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device=x.device)
        uc = cond if uc is None else uc
        x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
        return x, x.new_ones([x.shape[0]]), sigmas, len(sigmas), cond, uc

    def denoise(self, x, denoiser, sigma, cond, uc):
        denoised, _, _, rgb_list = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma), rgb_list

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0):
        denoised, rgb_list = self.denoise(x, denoiser, sigma, cond, uc)
        d = (x - denoised) / append_dims(sigma, x.ndim)
        return x + append_dims(next_sigma - sigma, x.ndim) * d, rgb_list

    def __call__(self, denoiser: Callable, x, cond: Dict, uc=None, num_steps=None, mask=None, init_im=None):
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        rgb_list = None
        for i in range(num_sigmas - 1):
            x, rgb_list = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc)
        return x, rgb_list

    forward = __call__


def cfg_euler_update(x: torch.Tensor, eps: torch.Tensor, sigma: torch.Tensor, sigma_next: torch.Tensor, scale: float, scale_im: float,
                     fused: bool = True) -> torch.Tensor:
    """One fused tail of a 3-way-CFG Euler step with EpsScaling: x [n,...] fp32, eps [3n,...] (u | ic | c) network output,
    sigma / sigma_next 0-d device tensors.  den_b = x - sigma eps_b; d0 = den_u + scale (den_c - den_ic) + scale_im (den_ic - den_u);
    x' = x + (x - d0) / sigma * (sigma_next - sigma)."""
    if fused and x.is_cuda:
        from . import ops
        return ops.cfg_euler_step(x, eps, sigma, sigma_next, scale, scale_im)
    e_u, e_ic, e_c = eps.float().chunk(3)
    den = [x - sigma * e for e in (e_u, e_ic, e_c)]
    d0 = den[0] + scale * (den[2] - den[1]) + scale_im * (den[1] - den[0])
    return x + (x - d0) / sigma * (sigma_next - sigma)


def fused_cfg3_euler_step(denoiser: "DiscreteDenoiser", network: Callable, x: torch.Tensor, sigma: torch.Tensor, sigma_next: torch.Tensor,
                          scale: float, scale_im: float, fused: bool = True) -> torch.Tensor:
    """ONE step of EulerEDMSampler.sampler_step (sampling.py:85-136) under ScheduledCFGImgTextRef (guiders.py:102-133) and
    DiscreteDenoiser + EpsScaling (denoiser.py:47-79), in the form the product's sampling job launches it (cd360/job.py):

        x3 = [x | x | x]                                   guider.prepare_inputs (the conditioning batch is assembled once per image)
        x_in, c_noise = c_in(sigma_q) x3, idx(sigma_q)     DiscreteDenoiser.network_inputs, sigma snapped to the table ON the device
        eps = network(x_in, c_noise)                       the UNet over the CFG batch
        x' = cfg_euler_update(x, eps, sigma, sigma_next)   c_out scaling + 3-way combine + to_d + Euler: cd360_cfg_euler_step_f32

    `network(x_in, c_noise) -> eps [3 n, ...]`; sigma / sigma_next 0-d device tensors of the sampler's schedule (table entries, so the
    snapped sigma_q of c_out equals the sigma of to_d, as in the reference's own run).  No host synchronisation anywhere in the step."""
    x3 = x.expand(3, *x.shape[1:]) if x.shape[0] == 1 else torch.cat([x] * 3)
    x_in, c_noise, _, _, _ = denoiser.network_inputs(x3, sigma.expand(x3.shape[0]), {})
    eps = network(x_in, c_noise)
    return cfg_euler_update(x, eps.contiguous(), sigma.reshape(1), sigma_next.reshape(1), scale, scale_im, fused=fused)
