"""Ray utilities of the pose path (reference sgm/modules/utils_cameraray.py:61-314) on the HIP kernels.

`pose` stays what the reference passes around -- a list (length b) of camera batches of size n+1, real pytorch3d
`PerspectiveCameras` or `cd360.cameras.PerspectiveCameras` -- and is packed once into a `[b, n+1, 16]` fp32 device
tensor; nothing ever goes back to the host (the reference bounces GPU->CPU->GPU per call, :79-99,191-193)."""
from __future__ import annotations

import torch

from cd360 import nerf as _nerf
from cd360 import ops
from cd360.cameras import pack_cameras


class PoseBuffer(list):
    """The camera batches of a forward (a plain list to every consumer: same objects, same identities -- the CFG de-duplication looks at
    those) together with the packed tensor the kernels read, owned by the caller.  A sampler whose steps are captured into hipGraphs
    points them at the next target pose with `rewrite(pose)`: the graphs keep reading the same buffer.  (Writing into the memoised
    packing of `packed_pose` instead would leave that memo entry describing cameras it no longer holds.)"""

    def __init__(self, pose, device):
        super().__init__(pose)
        self.packed = pack_cameras(pose, device)

    def rewrite(self, pose) -> None:
        new = pack_cameras(pose, self.packed.device)
        assert new.shape == self.packed.shape, "the captured graphs were sized for another camera layout"
        self.packed.copy_(new)


def packed_pose(pose, device) -> torch.Tensor:
    """Pack (and memoise on the list object's elements) the cameras of one forward call."""
    if isinstance(pose, PoseBuffer):
        # the buffer IS the pose (rewrite() changes it in place): never fall through to the memo over the original camera objects.
        # torch.device("cuda") != torch.device("cuda:0"), so compare type and resolved index
        want = torch.device(device)
        have = pose.packed.device
        same = want.type == have.type and (want.index is None or want.index == have.index)
        return pose.packed if same else pose.packed.to(want)
    if isinstance(pose, torch.Tensor):
        return pack_cameras(pose, device)
    # keyed on the camera objects AND on the storage / version of their fields: an in-place edit (cam.T = ..., the crop / scale adjusters
    # of data_co3d.py, a replaced list element) must not be served the stale packing
    def _fp(c):
        out = [id(c)]
        for name in ("R", "T", "focal_length", "principal_point"):
            t = getattr(c, name, None)
            out.append((t.data_ptr(), t._version) if isinstance(t, torch.Tensor) else None)
        return tuple(out)

    key = tuple(_fp(c) for c in pose)
    cache = _PACK_CACHE.get(key)
    if cache is not None and cache[0] == str(device):
        return cache[1]
    packed = pack_cameras(pose, device)
    if len(_PACK_CACHE) > 64:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (str(device), packed, pose)  # keep `pose` alive so ids stay unique
    return packed


_PACK_CACHE = {}


def get_patch_rays(cameras_list, num_patches_x, num_patches_y, device, return_xys=False, stratified=False):
    """(b, n+1, hw, 6) patch rays = (origin, unit direction), ray k = row*num_patches_x + col (:161-196)."""
    assert num_patches_x == num_patches_y, "square feature maps only (as the reference's reshape assumes)"
    r = num_patches_x
    cams = packed_pose(cameras_list, device)
    jx = torch.rand(r + 1) if stratified else None  # CPU RNG, x then y, as get_patch_raybundle draws them (:121-140)
    jy = torch.rand(r + 1) if stratified else None
    xs, ys = _nerf.patch_positions(r, device, jx), _nerf.patch_positions(r, device, jy)
    rays = ops.patch_rays(cams, xs, ys)
    if return_xys:
        hx, hy = torch.meshgrid(xs, ys, indexing="xy")
        return rays, torch.stack([hx.reshape(-1), hy.reshape(-1)], -1)[None]
    return rays


def get_plucker_parameterization(ray):
    """(d_hat, o x d_hat) (:201-219); tiny elementwise helper kept for API parity."""
    o, d = ray[..., :3], ray[..., 3:]
    d = d / d.norm(dim=-1).unsqueeze(-1)
    return torch.cat([d, torch.cross(o, d, dim=-1)], dim=-1)


def positional_encoding(ray, n_freqs=10, start_freq=0):
    """[sin(f_k x)]_k || [cos(f_k x)]_k with f_k = 2^(k - n/2) pi (:222-242)."""
    return _nerf.positional_encoding(ray, n_freqs)


# ---- novel-view camera paths used by sample.py:311-319 (reference :317-391), pytorch3d-free ----
def _translated_along_view_axis(cam1, axis: int, interp_start, interp_end, interp_step):
    """Slide the target camera's centre along one of its own VIEW axes: the new centre is the view-frame point
    `i * e_axis` mapped back to the world, the rotation is kept, and T is recomputed as `-C R` (reference :317-375).
    Steps follow `np.arange(start, end, step)` exactly as the reference does (float64 arange, then a float32 point)."""
    import numpy as np

    from cd360.cameras import PerspectiveCameras

    cameras = []
    view_to_world = cam1.get_world_to_view_transform().inverse()
    for i in np.arange(interp_start, interp_end, interp_step):
        p = np.zeros(3)
        p[axis] = i
        point = torch.from_numpy(p).reshape(1, 3).float().to(cam1.device)
        centre = view_to_world.transform_points(point)  # [1, 3]
        rt = cam1.R[0]
        new_t = -rt.T @ centre.T  # [3, 1]  ( = -(C R)^T )
        cameras.append(PerspectiveCameras(R=cam1.R, T=new_t.T, focal_length=cam1.focal_length, principal_point=cam1.principal_point,
                                          image_size=512, device=cam1.device))
    return cameras


def interpolate_translate_interpolate_xaxis(cam1, interp_start, interp_end, interp_step):
    return _translated_along_view_axis(cam1, 0, interp_start, interp_end, interp_step)


def interpolate_translate_interpolate_yaxis(cam1, interp_start, interp_end, interp_step):
    return _translated_along_view_axis(cam1, 1, interp_start, interp_end, interp_step)


def interpolate_translate_interpolate_zaxis(cam1, interp_start, interp_end, interp_step):
    return _translated_along_view_axis(cam1, 2, interp_start, interp_end, interp_step)


def interpolatefocal(cam1, interp_start, interp_end, interp_step):
    """Same pose, focal length scaled by each step of `np.arange(start, end, step)` (reference :378-391)."""
    import numpy as np

    from cd360.cameras import PerspectiveCameras

    return [PerspectiveCameras(R=cam1.R, T=cam1.T, focal_length=cam1.focal_length * i, principal_point=cam1.principal_point, image_size=512,
                               device=cam1.device) for i in np.arange(interp_start, interp_end, interp_step)]
