"""Fine-tuning objective of the pose path (reference sgm/modules/diffusionmodules/loss.py:108-215), YAML target
`sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef` (configs/train_co3d_concept.yaml:119-134).

`__call__` noises the target latent and the reference latents, runs the denoiser and hands everything to `get_loss`, which
returns the per-sample terms the engine weights (diffusion.py:226-241, see cd360.finetune.combine_losses):
  loss_l2  [b]          w(sigma) * (D(x) - x0)^2, mask-weighted mean
  loss_fg  [b, blocks]  (clamp(fg, 0, 1) - opacity_r)^2 per pose block, opacity resized (antialiased bilinear) to the block's r
  loss_bg  [b, blocks]  |alpha - opacity| (1 - opacity) on rays with opacity < 0.1
  loss_rgb [b, blocks]  masked (rgb_target_r - rgb_pred)^2 / sum(mask)
The arithmetic is a few small elementwise/resize ops on [b, 4, 64, 64]-sized tensors, done with torch on whatever device the
model outputs live -- except the three rendering terms of each pose block on the GPU, which are one kernel forward and one backward
(cd360_render_loss_f32: ~35 torch kernels per block otherwise, twelve blocks per step); the UNet forward it wraps is the HIP path, and
under autograd its backward runs on the HIP backward kernels (cd360/grad.py; cd360.finetune.train_step is one
optimisation step of config 4).
"""
import math
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from cd360 import ops

from ...util import append_dims, instantiate_from_config


class StandardDiffusionLossImgRef(nn.Module):
    def __init__(self, sigma_sampler_config: dict, sigma_sampler_config_ref: Optional[dict] = None, type: str = "l2",
                 offset_noise_level: float = 0.0, batch2model_keys: Optional[Union[str, List[str]]] = None):
        super().__init__()
        assert type in ["l2", "l1", "lpips"]
        if type == "lpips":
            raise NotImplementedError("the LPIPS network is outside the pose hot path (SURVEY.md §8); use l2 (the shipped config) or l1")
        self.sigma_sampler = instantiate_from_config(sigma_sampler_config)
        self.sigma_sampler_ref = instantiate_from_config(sigma_sampler_config_ref) if sigma_sampler_config_ref is not None else None
        self.type = type
        self.offset_noise_level = offset_noise_level
        if not batch2model_keys:
            batch2model_keys = []
        if isinstance(batch2model_keys, str):
            batch2model_keys = [batch2model_keys]
        self.batch2model_keys = set(batch2model_keys)

    def _noise_like(self, x):
        noise = torch.randn_like(x)
        if self.offset_noise_level > 0.0:
            noise = noise + self.offset_noise_level * append_dims(torch.randn(x.shape[0], device=x.device), x.ndim)
        return noise

    def __call__(self, network, denoiser, conditioner, input, input_rgb, input_ref, pose, mask, mask_ref, opacity, batch):
        """(:138-176) same draw order as the reference: sigma, noise, [offset], sigma_ref, noise_ref, [offset]."""
        cond = conditioner(batch)
        extra = {key: batch[key] for key in self.batch2model_keys.intersection(batch)}
        sigmas = self.sigma_sampler(input.shape[0]).to(input.device)
        noise = self._noise_like(input)
        extra["pose"] = pose
        extra["mask_ref"] = mask_ref
        noised_input = input + noise * append_dims(sigmas, input.ndim)
        if self.sigma_sampler_ref is not None:
            sigmas_ref = self.sigma_sampler_ref(input.shape[0]).to(input.device)
            if input_ref is not None:
                input_ref = input_ref + self._noise_like(input_ref) * append_dims(sigmas_ref, input_ref.ndim)
            extra["sigmas_ref"] = sigmas_ref
        extra["input_ref"] = input_ref
        model_output, fg_mask_list, alphas, predicted_rgb_list = denoiser(network, noised_input, sigmas, cond, **extra)
        w = append_dims(denoiser.w(sigmas), input.ndim)
        return self.get_loss(model_output, fg_mask_list, predicted_rgb_list, input, input_rgb, w, mask, mask_ref, opacity, alphas)

    def get_loss(self, model_output, fg_mask_list, predicted_rgb_list, target, target_rgb, w, mask, mask_ref, opacity, alphas_list):
        """(:178-209 for l2, :210-213 for l1).  All terms in fp32."""
        f32 = lambda t: None if t is None else t.float()
        model_output, target, target_rgb, w, mask, opacity = map(f32, (model_output, target, target_rgb, w, mask, opacity))
        loss_rgb, loss_fg, loss_bg = [], [], []
        if self.type == "l1":
            return torch.mean((w * (model_output - target).abs()).reshape(target.shape[0], -1), 1), loss_rgb
        loss = w * (model_output - target) ** 2
        if mask is not None:
            loss_l2 = (loss * mask).sum([1, 2, 3]) / (mask.sum([1, 2, 3]) + 1e-6)
        else:
            loss_l2 = torch.mean(loss.reshape(target.shape[0], -1), 1)
        n_blk = len(fg_mask_list)
        if (n_blk > 0 and n_blk == len(alphas_list) and len(predicted_rgb_list) in (0, n_blk) and (mask is not None or not predicted_rgb_list)
                and all(fg_mask.size(1) == rgb.size(1) for fg_mask, rgb in zip(fg_mask_list, predicted_rgb_list))
                and ops.render_loss_ok(fg_mask_list[0], alphas_list[0], predicted_rgb_list[0] if predicted_rgb_list else None, opacity, mask, target_rgb)):
            # GPU: the three terms of a block in one kernel (and one for their gradients); the resized targets are shared as below
            terms, resized, bg_w = [], {}, None
            mask_den = mask.sum([1, 2, 3]) + 1e-6 if predicted_rgb_list else None
            for i, (fg_mask, alphas) in enumerate(zip(fg_mask_list, alphas_list)):
                size = int(math.sqrt(fg_mask.size(1)))
                if opacity.shape[-2:] != (size, size):
                    opacity = F.interpolate(opacity, size=size, antialias=True, mode="bilinear").detach()
                    bg_w = None
                op = opacity.reshape(-1, size * size)
                if bg_w is None:
                    bg_w = ((1 - op) * ((op < 0.1) * 1)).contiguous()
                rgb = mask_ = want = None
                if predicted_rgb_list:
                    if size not in resized:
                        resized[size] = (F.interpolate(mask, size=size, antialias=True, mode="bilinear").detach().contiguous(),
                                         F.interpolate(target_rgb * 0.5 + 0.5, size=size, antialias=True, mode="bilinear").detach().contiguous())
                    (mask_, want), rgb = resized[size], predicted_rgb_list[i]
                terms.append(ops.render_loss(fg_mask, alphas, rgb, op, bg_w, mask_, want, mask_den))
            terms = torch.stack(terms, 1)  # [b, blocks, 3]
            return loss_l2, terms[..., 0], terms[..., 1], (terms[..., 2] if predicted_rgb_list else loss_rgb)
        if len(fg_mask_list) > 0 and len(alphas_list) > 0:
            bg_w = None  # (1 - opacity) * [opacity < 0.1] of the current map: constant until the next resize (the factor [..] is 0 / 1, so
            for fg_mask, alphas in zip(fg_mask_list, alphas_list):  # folding it into the weight first changes no bit of the product)
                size = int(math.sqrt(fg_mask.size(1)))
                # as in the reference, `opacity` is re-assigned: every block resizes the PREVIOUS block's resized map.  A resize to the
                # size the map already has is the identity (the antialias filter at scale 1 has the weights 1, 0): not launched
                if opacity.shape[-2:] != (size, size):
                    opacity = F.interpolate(opacity, size=size, antialias=True, mode="bilinear").detach()
                    bg_w = None
                op = opacity.reshape(-1, size * size)
                fg = torch.clamp(fg_mask.float().reshape(-1, size * size), 0.0, 1.0)
                loss_fg.append(((fg - op) ** 2).mean(1))
                op4 = op.reshape(-1, size * size, 1, 1)
                if bg_w is None:
                    bg_w = (1 - op4) * ((op4 < 0.1) * 1)
                loss_bg.append(((alphas.float() - op4).abs() * bg_w).mean([1, 2, 3]))
            loss_fg, loss_bg = torch.stack(loss_fg, 1), torch.stack(loss_bg, 1)
        if len(predicted_rgb_list) > 0:
            resized = {}  # per feature-grid size: the blocks of one resolution share the resized mask / rgb target
            mask_den = mask.sum([1, 2, 3]) + 1e-6
            for rgb in predicted_rgb_list:
                size = int(math.sqrt(rgb.size(1)))
                if size not in resized:
                    resized[size] = (F.interpolate(mask, size=size, antialias=True, mode="bilinear").detach(),
                                     F.interpolate(target_rgb * 0.5 + 0.5, size=size, antialias=True, mode="bilinear").detach())
                mask_, want = resized[size]
                err = (want - rgb.float().reshape(-1, size, size, 3).permute(0, 3, 1, 2)) ** 2
                loss_rgb.append((err * mask_).sum([1, 2, 3]) / mask_den)
            loss_rgb = torch.stack(loss_rgb, 1)
        return loss_l2, loss_fg, loss_bg, loss_rgb
