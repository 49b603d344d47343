"""FeatureNeRF module of the pose-conditioned blocks (reference sgm/modules/nerfsd_pytorch3d.py), HIP-backed.

Same classes, constructor arguments, return tuples and state_dict names as the reference:
  FeatureNeRFEncoding  plane_coefs.{0,2}.{weight,bias}, nviews.{weight,bias}, decoder.weight        (:23-51)
  VolRender            (no parameters)                                                               (:164-231)
  Raymarcher           buffers u, lengths, lengths_center, lengths_upper, lengths_lower              (:234-262)
  NerfSDModule         raymarcher.*, model.*                                                          (:397-464)
pytorch3d is not needed: cameras are packed tensors and all ray / projection math is in the kernels.
Importance sampling (`prev_weights`, Raymarcher.importance_sampling, pytorch3d._C.sample_pdf) is dead code in the
reference and is not provided: a non-None `prev_weights` raises.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from cd360 import nerf as _nerf
from cd360 import ops
from ..modules.diffusionmodules.util import zero_module
from ..modules.utils_cameraray import packed_pose


class FeatureNeRFEncoding(nn.Module):
    def __init__(self, in_channels, out_channels, far_plane: float = 2.0, rgb_predict=False, average=False, num_freqs=16) -> None:
        super().__init__()
        self.far_plane, self.rgb_predict, self.average, self.num_freqs = far_plane, rgb_predict, average, num_freqs
        if num_freqs != 16:
            raise NotImplementedError("the fused kernel is specialised for num_freqs=16 (the only value the reference uses)")
        dim = 3
        e = in_channels + num_freqs * dim * 4 + 2 * dim
        self.plane_coefs = nn.Sequential(nn.Linear(e, out_channels), nn.SiLU(), nn.Linear(out_channels, out_channels))
        self.nviews = nn.Linear(e, 1)
        self.decoder = zero_module(nn.Linear(out_channels, 1 + (3 if rgb_predict else 0), bias=False))
        self._fused = None

    def fused_weights(self) -> _nerf.FusedNerfWeights:
        ps = [self.plane_coefs[0].weight, self.plane_coefs[0].bias, self.plane_coefs[2].weight, self.plane_coefs[2].bias,
              self.nviews.weight, self.nviews.bias, self.decoder.weight]
        key = tuple((p.data_ptr(), p._version, p.device, p.dtype) for p in ps)
        if self.average:
            # nerfsd_pytorch3d.py:156-158: the views are averaged instead of softmax-weighted.  A zero `nviews` makes every view logit 0,
            # the in-kernel view softmax uniform = the mean (and, as in the reference, nviews then receives no gradient)
            ps[4], ps[5] = torch.zeros_like(ps[4]), torch.zeros_like(ps[5])
        if torch.is_grad_enabled() and any(p.requires_grad for p in ps):  # training: derived weights stay on the autograd tape
            wd = self.decoder.weight
            if wd.shape[0] == 1:
                wd = torch.cat([torch.zeros(3, wd.shape[1], device=wd.device, dtype=wd.dtype), wd], 0)
            return _nerf.FusedNerfWeights(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], wd, live=True)
        if self._fused is None or self._fused[0] != key:
            wd = self.decoder.weight
            if wd.shape[0] == 1:  # rgb_predict False: only sigma; pad rgb rows with zeros (row 3 = sigma)
                wd = torch.cat([torch.zeros(3, wd.shape[1], device=wd.device, dtype=wd.dtype), wd], 0)
            self._fused = (key, _nerf.FusedNerfWeights(ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], wd))
        return self._fused[1]

    def forward(self, pose, xref, ray_points, rays, mask_ref):
        raise NotImplementedError(
            "FeatureNeRFEncoding is evaluated through NerfSDModule.forward (rays and sample points never leave the kernel); "
            "the stand-alone (ray_points, rays) call form of the reference is not provided")


class VolRender(nn.Module):
    """forward(features [b,hw,S,C], densities [b,hw,S,1] (already exp'ed), dists [*,S,1], ..., rgb (already sigmoid'ed))
    -> (rendered, fg_mask, alphas, weights|weights_uniform|None, rgb)   (:196-231)."""

    def get_weights(self, densities, deltas):
        b, hw, S = densities.shape[:3]
        dummy = torch.zeros(b, hw, S, 4, dtype=torch.float32, device=densities.device)
        _, _, alphas, weights, _ = ops.volrender(dummy, densities.reshape(b, hw, S), _dists2d(deltas, hw, S), None, want_weights=True,
                                                 sigma_is_raw=False)
        return weights, alphas, torch.exp(-(torch.cumsum(deltas * densities, -2) - deltas * densities))

    def forward(self, features, densities, dists=None, return_weight=False, densities_uniform=None, dists_uniform=None,
                return_weights_uniform=False, rgb=None):
        if dists is None:
            raise NotImplementedError("direct-weights mode (dists=None) is unused by the reference's pose path")
        if densities_uniform is not None:
            raise NotImplementedError("importance sampling is dead code in the reference (SURVEY.md F3)")
        b, hw, S, C = features.shape
        rendered, fg, alphas, weights, rgb_out = ops.volrender(
            features, densities.reshape(b, hw, S), _dists2d(dists, hw, S), None if rgb is None else rgb, want_weights=return_weight,
            sigma_is_raw=False, rgb_is_raw=False)
        if return_weight:
            return rendered, fg, alphas, weights, rgb_out
        return rendered, fg, alphas, None, rgb_out


def _dists2d(dists: torch.Tensor, hw: int, S: int) -> torch.Tensor:
    """[1|b, hw|1, S, 1] (or [S] / [hw, S]) -> [S] or [hw, S]; the reference shares dists across the batch."""
    d = dists.reshape(-1, S) if dists.dim() <= 2 else dists.reshape(dists.shape[0], -1, S)[0]
    d = d.float()
    return d[0].contiguous() if d.shape[0] == 1 else d.reshape(hw, S).contiguous()


class Raymarcher(nn.Module):
    def __init__(self, num_samples=32, far_plane=2.0, stratified=False, training=True, imp_sampling_percent=0.9, near_plane=0.0):
        super().__init__()
        self.num_samples, self.far_plane, self.near_plane = num_samples, far_plane, near_plane
        u_max = 1.0 / num_samples
        self.register_buffer("u", torch.linspace(0, 1 - u_max, num_samples))
        lengths = torch.linspace(near_plane, near_plane + far_plane, num_samples + 1)
        center = (lengths[..., 1:] + lengths[..., :-1]) / 2.0
        self.register_buffer("lengths", lengths)
        self.register_buffer("lengths_center", center)
        self.register_buffer("lengths_upper", torch.cat([center, lengths[..., -1:]], -1))
        self.register_buffer("lengths_lower", torch.cat([lengths[..., :1], center], -1))
        self.stratified = stratified
        self.training = training
        self.imp_sampling_percent = imp_sampling_percent
        self.device_rng = False  # True: patch x / y jitter drawn on the device generator too (no host tensor in the step: hipGraph capture)

    def jitter(self, resolution: int, device):
        """The three uniform draws of a stratified training step, in the reference's order and on the reference's
        generators: patch x then y on the CPU RNG (utils_cameraray.py:121-140), depths on the device RNG (:317-325)."""
        if not (self.stratified and self.training):
            return None, None
        if self.device_rng:
            jx, jy = torch.rand(resolution + 1, device=device), torch.rand(resolution + 1, device=device)
        else:
            jx, jy = torch.rand(resolution + 1), torch.rand(resolution + 1)
        jd = torch.rand((resolution ** 2, self.num_samples + 1), dtype=torch.float32, device=device)
        return (jx, jy), jd

    @torch.no_grad()
    def forward(self, pose, resolution, weights, imp_sample_next_step=False, device="cuda", pytorch3d=True):
        """-> (rays [b,n+1,hw,6], ray_points [b,1,hw,S,3], dists [1,hw,S], None, None)   (:332-394)"""
        if weights is not None:
            raise NotImplementedError("importance sampling is dead code in the reference (SURVEY.md F3)")
        cams = packed_pose(pose, device)
        xy, jd = self.jitter(resolution, device)
        xs = _nerf.patch_positions(resolution, device, None if xy is None else xy[0])
        ys = _nerf.patch_positions(resolution, device, None if xy is None else xy[1])
        t, dists = _nerf.depth_samples(self.num_samples, self.far_plane - self.near_plane, self.near_plane, device, resolution ** 2, jd)
        rays = ops.patch_rays(cams, xs, ys)
        pts = ops.ray_project_index(cams, xs, ys, t, want_grid=False, want_index=False)["points"]
        hw = resolution ** 2
        dists = dists[None].expand(hw, -1) if dists.dim() == 1 else dists
        return rays, pts[:, None], dists[None], None, None


class NerfSDModule(nn.Module):
    def __init__(self, mode="feature-nerf", out_channels=None, far_plane=2.0, num_samples=32, rgb_predict=False, average=False,
                 num_freqs=16, stratified=False, imp_sampling_percent=0.9, near_plane=0.0):
        super().__init__()
        if mode != "feature-nerf":
            raise KeyError(mode)
        self.rgb_predict = rgb_predict
        self.far, self.near, self.num_samples = far_plane, near_plane, num_samples
        self.raymarcher = Raymarcher(num_samples=num_samples, far_plane=near_plane + far_plane, stratified=stratified,
                                     imp_sampling_percent=imp_sampling_percent, near_plane=near_plane)
        self.model = FeatureNeRFEncoding(out_channels, out_channels, far_plane=near_plane + far_plane, rgb_predict=rgb_predict,
                                         average=average, num_freqs=num_freqs)
        self.return_view_weights = True  # the reference returns plane_features_attn; its only caller drops it

    def render_inputs(self, pose, xref, mask_ref=None, tables=None, want_view_weights=False, dims=None):
        """Fast path used by BasicTransformerBlock: -> (h [b,hw,S,C] bf16, dec [b,hw,S,4] fp32 = (rgb_raw 0..2, sigma_raw 3),
        dists [S]|[hw,S], view_weights|None)."""
        if xref is not None and xref.dim() == 5:
            xref = xref.reshape(*xref.shape[:2], -1, xref.shape[-1])
        b, n, hw, C = xref.shape if xref is not None else dims
        device = xref.device if xref is not None else tables[0].device
        if mask_ref is not None:  # nerfsd_pytorch3d.py:61-70
            r = int(math.isqrt(hw))
            m = torch.nn.functional.interpolate(mask_ref.reshape(b * n, *mask_ref.shape[2:]).float(), size=[r, r], mode="nearest")
            xref = xref * m.reshape(b, n, -1, 1).to(xref.dtype)
            tables = None
        cams = packed_pose(pose, device)
        xy, jd = self.raymarcher.jitter(int(math.isqrt(hw)), device)
        return _nerf.fused_feature_nerf(self.model.fused_weights(), cams, xref, self.num_samples, self.far, self.near, xy, jd,
                                        want_view_weights, tables, (b, n, hw, C))

    def forward(self, pose, xref=None, mask_ref=None, prev_weights=None, imp_sample_next_step=False):
        """-> (features [b,hw,S,C], sigma_raw [b,hw,S,1], dists [1,hw,S,1], view_weights [b,n,hw,S,1], rgb_raw [b,hw,S,3]|None,
        None, None)   (:434-464)"""
        if prev_weights is not None:
            raise NotImplementedError("importance sampling is dead code in the reference (SURVEY.md F3)")
        h, dec, dists, vw = self.render_inputs(pose, xref, mask_ref, want_view_weights=self.return_view_weights and not self.model.average)
        hw, S = h.shape[1], h.shape[2]
        d = dists[None].expand(hw, -1) if dists.dim() == 1 else dists
        rgb = dec[..., :3] if self.rgb_predict else None
        return h, dec[..., 3:], d[None, :, :, None], vw, rgb, None, None
