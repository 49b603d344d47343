"""FeatureNeRF module of the pose-conditioned blocks (reference sgm/modules/nerfsd_pytorch3d.py), HIP-backed.

Same classes, constructor arguments, return tuples and state_dict names as the reference:
  FeatureNeRFEncoding  plane_coefs.{0,2}.{weight,bias}, nviews.{weight,bias}, decoder.weight        (:23-51)
  VolRender            (no parameters)                                                               (:164-231)
  Raymarcher           buffers u, lengths, lengths_center, lengths_upper, lengths_lower              (:234-262)
  NerfSDModule         raymarcher.*, model.*                                                          (:397-464)
pytorch3d is not needed: cameras are packed tensors and all ray / projection math is in the kernels.
Importance sampling (`prev_weights`, Raymarcher.importance_sampling, pytorch3d._C.sample_pdf) is dead
code in the reference: NerfSDModule.forward never forwards `imp_sample_next_step` to the raymarcher (:442 vs :333), so no block ever
receives weights to sample from.  It is provided here in the form the code evidently intends: a non-None `prev_weights`
draws the sample depths by inverse-CDF sampling (cd360_sample_pdf in place of pytorch3d's C op) per batch element and ray, and
`NerfSDModule.honour_imp_sample_next_step = True` (default False = the reference's behaviour: the argument is accepted and dropped)
adds the no-grad density pass at the uniform depths that feeds the next block's `prev_weights`.
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
        b, hw, S, C = features.shape
        per_batch = _dists_per_batch(dists, b, hw, S)
        if per_batch is None:
            rendered, fg, alphas, weights, rgb_out = ops.volrender(
                features, densities.reshape(b, hw, S), _dists2d(dists, hw, S), None if rgb is None else rgb, want_weights=return_weight,
                sigma_is_raw=False, rgb_is_raw=False)
        else:  # importance-sampled depths differ per batch element (f4): the kernel shares dists across its batch, so one launch each
            parts = [ops.volrender(features[i:i + 1], densities[i:i + 1].reshape(1, hw, S), per_batch[i],
                                   None if rgb is None else rgb[i:i + 1], want_weights=return_weight, sigma_is_raw=False, rgb_is_raw=False)
                     for i in range(b)]
            rendered, fg, alphas, weights, rgb_out = (None if parts[0][j] is None else torch.cat([p_[j] for p_ in parts], 0) for j in range(5))
        if return_weight:
            return rendered, fg, alphas, weights, rgb_out
        if return_weights_uniform and densities_uniform is not None:  # :227-229
            return rendered, fg, alphas, self.get_weights(densities_uniform, dists_uniform)[0], rgb_out
        return rendered, fg, alphas, None, rgb_out


def _dists_per_batch(dists: torch.Tensor, b: int, hw: int, S: int):
    """None when one set of dists serves the whole batch (every path of the reference but importance sampling); else b tensors [hw, S]."""
    if dists.dim() < 3 or dists.shape[0] == 1 or b == 1:
        return None
    d = dists.reshape(dists.shape[0], -1, S).float()
    assert d.shape[0] == b, "dists carry a batch dimension that is not the features'"
    return [d[i].expand(hw, S).contiguous() for i in range(b)]


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
    def importance_sampling(self, cdf, num_rays, num_samples, device):
        """(:264-306) cdf [b, hw', S, 1]: the previous block's rendering weights at the uniform depths -> (lengths [b, num_rays, S],
        dists [b, num_rays, S]) by inverse-CDF sampling.  The weight maps are resized to this block's ray grid if needed (:269-286), a
        row that sums to less than 1e-5 is padded up to it (:290-293), and ops.sample_pdf (cd360_sample_pdf) stands where
        pytorch3d._C.sample_pdf does.  As the reference is written `u` is an expanded view, so its reshape hands the C op a copy and
        the samples are lost (and the stratified `u +=` raises): the evident intent is implemented -- the outputs of sample_pdf are
        the returned lengths.  The stratified draw (:296-298) comes from the device generator."""
        cdf = cdf[..., 0].float() + 0.01
        if cdf.shape[1] != num_rays:
            size, size_ = int(math.isqrt(num_rays)), int(math.isqrt(cdf.size(1)))
            m = cdf.permute(0, 2, 1).reshape(cdf.shape[0], -1, size_, size_)
            m = torch.nn.functional.interpolate(m, size=[size, size], antialias=True, mode="bilinear")
            cdf = m.reshape(cdf.shape[0], -1, size * size).permute(0, 2, 1)
        cdf = cdf.to(device)
        b = cdf.shape[0]
        # the buffers follow the module's dtype (bf16 after .to(bf16)): the fp32 grids are rebuilt from the scalars, on the host, once
        key = (int(num_samples), float(self.near_plane), float(self.far_plane), str(device))
        lengths_, u_ = _nerf._const(("imp_grids",) + key, lambda: (
            torch.linspace(self.near_plane, self.near_plane + self.far_plane, num_samples + 1).to(device),
            torch.linspace(0, 1 - 1.0 / num_samples, num_samples).to(device)))
        lengths = lengths_[None, None, :].expand(b, num_rays, -1)
        cdf_sum = torch.sum(cdf, dim=-1, keepdim=True)
        padding = torch.relu(1e-5 - cdf_sum)
        cdf = cdf + padding / cdf.shape[-1]
        pdf = cdf / (cdf_sum + padding)
        u_max = 1.0 / num_samples
        u = u_[None, None, :].expand(b, num_rays, -1)
        if self.stratified and self.training:
            u = u + torch.rand((b, num_rays, num_samples), dtype=cdf.dtype, device=device) * u_max
        return ops.sample_pdf(lengths, pdf, u, 1e-5, want_dists=True)

    def _depths(self, resolution, weights, device, jd):
        """The branch of :345-355: stratified / uniform depths shared by the batch, or importance-sampled ones per batch element."""
        hw = resolution ** 2
        if weights is not None and self.imp_sampling_percent > 0:
            coin = torch.rand(1)  # drawn on the CPU generator in every mode, as :348 does
            if not (bool(coin < (1.0 - self.imp_sampling_percent)) and self.training):
                return self.importance_sampling(weights, hw, self.num_samples, device)
        return _nerf.depth_samples(self.num_samples, self.far_plane - self.near_plane, self.near_plane, device, hw, jd)

    @torch.no_grad()
    def forward(self, pose, resolution, weights, imp_sample_next_step=False, device="cuda", pytorch3d=True):
        """-> (rays [b,n+1,hw,6], ray_points [b,1,hw,S,3], dists [1|b,hw,S], ray_points_uniform|None, dists_uniform|None)   (:332-394)"""
        cams = packed_pose(pose, device)
        xy, jd = self.jitter(resolution, device)
        xs = _nerf.patch_positions(resolution, device, None if xy is None else xy[0])
        ys = _nerf.patch_positions(resolution, device, None if xy is None else xy[1])
        t, dists = self._depths(resolution, weights, device, jd)
        rays = ops.patch_rays(cams, xs, ys)
        hw = resolution ** 2

        def points(tt):
            if tt.dim() == 3:  # per batch element
                return torch.cat([ops.ray_project_index(cams[i:i + 1], xs, ys, tt[i].contiguous(), want_grid=False, want_index=False)["points"]
                                  for i in range(tt.shape[0])], 0)
            return ops.ray_project_index(cams, xs, ys, tt, want_grid=False, want_index=False)["points"]

        pts = points(t)
        pts_u = du = None
        if imp_sample_next_step:  # :357-371
            tu, du = _nerf.depth_samples(self.num_samples, self.far_plane - self.near_plane, self.near_plane, device, hw, None)
            pts_u, du = points(tu)[:, None], du[None].expand(hw, -1)[None]
        if dists.dim() == 3:
            return rays, pts[:, None], dists, pts_u, du
        dists = dists[None].expand(hw, -1) if dists.dim() == 1 else dists
        return rays, pts[:, None], dists[None], pts_u, du


def _tables_of(tables, i: int, n: int):
    """The reference tables (Y, lv[, img_map]) as batch element i alone sees them."""
    if tables is None:
        return None
    Y, lv = tables[0], tables[1]
    im = tables[2] if len(tables) > 2 else None
    if im is None:
        return Y[i * n:(i + 1) * n], lv[i * n:(i + 1) * n]
    return Y, lv, im[i * n:(i + 1) * n].contiguous()


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
        # False = the reference: `imp_sample_next_step` is accepted and dropped (:442 never forwards it).  True = what the argument is for
        self.honour_imp_sample_next_step = False

    def render_inputs(self, pose, xref, mask_ref=None, tables=None, want_view_weights=False, dims=None, prev_weights=None,
                      uniform_depths=False):
        """Fast path used by BasicTransformerBlock: -> (h [b,hw,S,C] bf16, dec [b,hw,S,4] fp32 = (rgb_raw 0..2, sigma_raw 3),
        dists [S]|[hw,S] (|[b,hw,S] importance-sampled), view_weights|None).  prev_weights: depths by Raymarcher.importance_sampling
        (one render per batch element: the kernels share the depths of a launch across its batch); uniform_depths: no jitter."""
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
        r = int(math.isqrt(hw))
        xy, jd = (None, None) if uniform_depths else self.raymarcher.jitter(r, device)
        fw = self.model.fused_weights()
        if prev_weights is not None and not uniform_depths:
            t, dists = self.raymarcher._depths(r, prev_weights, device, jd)
            if t.dim() == 3:
                parts = [_nerf.fused_feature_nerf(fw, cams[i:i + 1], None if xref is None else xref[i:i + 1], self.num_samples, self.far,
                                                  self.near, xy, None, want_view_weights,
                                                  _tables_of(tables, i, n),
                                                  (1, n, hw, C), depths=(t[i].contiguous(), dists[i].contiguous())) for i in range(b)]
                cat = lambda j: None if parts[0][j] is None else torch.cat([p_[j] for p_ in parts], 0)
                return cat(0), cat(1), dists, cat(3)
            return _nerf.fused_feature_nerf(fw, cams, xref, self.num_samples, self.far, self.near, xy, None, want_view_weights, tables,
                                            (b, n, hw, C), depths=(t, dists))
        return _nerf.fused_feature_nerf(fw, cams, xref, self.num_samples, self.far, self.near, xy, jd, want_view_weights, tables, (b, n, hw, C))

    def forward(self, pose, xref=None, mask_ref=None, prev_weights=None, imp_sample_next_step=False):
        """-> (features [b,hw,S,C], sigma_raw [b,hw,S,1], dists [1|b,hw,S,1], view_weights [b,n,hw,S,1], rgb_raw [b,hw,S,3]|None,
        sigma_raw_uniform|None, dists_uniform|None)   (:434-464).  The last two are None unless `honour_imp_sample_next_step` is set
        and imp_sample_next_step is True (the reference always returns None there: see the module docstring)."""
        h, dec, dists, vw = self.render_inputs(pose, xref, mask_ref, want_view_weights=self.return_view_weights and not self.model.average,
                                               prev_weights=prev_weights)
        hw, S = h.shape[1], h.shape[2]
        rgb = dec[..., :3] if self.rgb_predict else None
        sig_u = d_u = None
        if imp_sample_next_step and self.honour_imp_sample_next_step:  # :453-457
            with torch.no_grad():
                _, dec_u, du, _ = self.render_inputs(pose, xref, mask_ref, uniform_depths=True)
                sig_u = dec_u[..., 3:]
                d_u = (du[None].expand(hw, -1) if du.dim() == 1 else du)[None, :, :, None]
        if dists.dim() == 3:
            return h, dec[..., 3:], dists[..., None], vw, rgb, sig_u, d_u
        d = dists[None].expand(hw, -1) if dists.dim() == 1 else dists
        return h, dec[..., 3:], d[None, :, :, None], vw, rgb, sig_u, d_u
