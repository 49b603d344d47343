"""Camera normalisation and crop/scale intrinsics of the reference's CO3D loader, pytorch3d-free.

Same functions, arguments and return tuples as sgm/data/data_co3d.py:27-185 (intersect_skew_line_groups,
intersect_skew_lines_high_dim, compute_optical_axis_intersection, normalize_cameras, centerandalign, square_bbox) plus the two
pytorch3d.implicitron.dataset.utils helpers the loader calls at :458-467 (adjust_camera_to_bbox_crop_,
adjust_camera_to_image_scale_) -- written from the geometry, on the packed `[R | T | f | pp]` rows the rest of this package works on,
not from the reference's statements:

  * a camera row gives its centre `c = -T R^T` and its optical axis `a = R[:, 2]` directly (the unprojection of the principal point at
    depth 1 is `c + a`: the reference reaches the same two values through `unproject_points` and a transform inverse);
  * the point nearest to all axes solves the 3 x 3 normal equations `(sum_i w_i (I - a_i a_i^T)) x = sum_i w_i (c_i - a_i (a_i . c_i))`
    in closed form (cross-product adjugate, float64) -- the reference builds one projector per line and calls `lstsq`;
  * moving the world origin to that point is `T' = x R + T` row by row (no 4 x 4 transforms), aligning the mean up vector `mean_i
    R_i[:, 1]` with +Y is `R' = Q^T R` with the closed-form rotation `Q = c I + [v]_x + v v^T / (1 + c)`.

The golden vectors of tests/golden/cameras.npz -- produced by the REFERENCE's functions compiled in place -- pin every return value
(tests/test_cameras_cpu.py, tests/test_f_rows_gpu.py).  pytorch3d itself is an un-pinned third-party dependency that is not in the
reference tree; its NDC<->pixel conversion is restated from the published algorithm (min-side-normalised NDC:
p_px = half_size - p_ndc * min(half_size), f_px = f_ndc * min(half_size)) and is pinned here only by known-answer tests.

These run once per dataset on a few hundred cameras: host torch on CPU, like the reference.  They produce the rows that
cd360.cameras.pack_cameras hands to the HIP path.
"""
from __future__ import annotations

import numpy as np
import torch

from cd360.cameras import PerspectiveCameras, join_cameras_as_batch


# ---- lines in space ----
def _solve_sym3(A: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """x with A x = b for symmetric 3 x 3 A [..., 3, 3], b [..., 3]: rows of the adjugate are cross products of A's rows."""
    r0, r1, r2 = A[..., 0, :], A[..., 1, :], A[..., 2, :]
    c12, c20, c01 = torch.linalg.cross(r1, r2), torch.linalg.cross(r2, r0), torch.linalg.cross(r0, r1)
    det = (r0 * c12).sum(-1, keepdim=True)
    return torch.stack([(c12 * b).sum(-1), (c20 * b).sum(-1), (c01 * b).sum(-1)], -1) / det


def intersect_skew_lines_high_dim(p, r, mask=None):
    """Least-squares point nearest to the lines `p + t r` of the second-to-last axis: p, r [..., n, dim], mask [..., n] line weights
    (:39-55).  Returns (point [..., dim], the normalised directions).  Raises AssertionError on a degenerate bundle, as the reference."""
    dirs = torch.nn.functional.normalize(r, dim=-1)
    w = torch.ones_like(p[..., 0]) if mask is None else mask
    d64, p64, w64 = dirs.double(), p.double(), w.double()
    dim = p.shape[-1]
    # sum_i w_i (I - d_i d_i^T)  and  sum_i w_i (p_i - d_i (d_i . p_i))
    normal = w64.sum(-1)[..., None, None] * torch.eye(dim, dtype=torch.float64, device=p.device) - torch.einsum("...n,...ni,...nj->...ij", w64, d64, d64)
    rhs = (w64[..., None] * (p64 - d64 * (d64 * p64).sum(-1, keepdim=True))).sum(-2)
    # A rank-deficient bundle (one camera; all optical axes parallel) has no unique nearest point; the reference's lstsq (gelsy) then
    # returns the minimum-norm one and goes on -- so does this (the closed form would divide by a vanishing determinant).
    scale = normal.diagonal(dim1=-2, dim2=-1).abs().amax(-1).clamp_min(1e-300)
    well = torch.linalg.det(normal).abs() > 1e-10 * scale ** dim
    if dim == 3 and bool(well.all()):
        x = _solve_sym3(normal, rhs)
    else:
        x = torch.linalg.lstsq(normal, rhs[..., None], driver="gelsy" if normal.device.type == "cpu" else None).solution[..., 0]
    if not torch.isfinite(x).all():
        raise AssertionError(f"degenerate line bundle: no unique nearest point ({x})")
    return x.to(p.dtype), dirs


def _foot_on_line(origin, direction, point):
    """Foot of the perpendicular from `point` on the line origin + t direction (unit direction), and its length."""
    along = ((point - origin) * direction).sum(-1, keepdim=True)
    foot = origin + along * direction
    return (point - foot).norm(dim=-1), foot


def _point_line_distance(p1, r1, p2):
    """(distance of p2 from the line p1 + t r1, foot of the perpendicular) -- the reference's helper name (:58-63)."""
    return _foot_on_line(p1, r1, p2)


def intersect_skew_line_groups(p, r, mask):
    """(:27-36) -> (nearest point, its foot on every line, squared distances to the lines, normalised directions)."""
    x, dirs = intersect_skew_lines_high_dim(p, r, mask=mask)
    target = x[..., None, :]
    _, feet = _foot_on_line(p, dirs, target)
    return x, feet, ((feet - target) ** 2).sum(-1), dirs


# ---- the rig ----
def _centres_and_axes(cameras):
    """Per camera row: centre c = -T R^T and optical axis R[:, 2] (view-space +Z in world coordinates)."""
    R, T = cameras.R.float(), cameras.T.float()
    return -(T[:, None, :] * R).sum(-1), R[:, :, 2]


def compute_optical_axis_intersection(cameras):
    """(:66-91) -> (p_intersect [1, 3], distance of every camera centre from it [1, 1, n], feet on the axes [1, 1, n, 3], the principal
    points unprojected at depth 1 [n, 3], unit axis directions [1, 1, n, 3])."""
    centres, axes = _centres_and_axes(cameras)
    x, feet, _, dirs = intersect_skew_line_groups(centres[None, None], axes[None, None], None)
    x = x.reshape(1, 3)
    return x, (x - centres[None, None]).norm(dim=-1), feet, centres + axes, dirs


def normalize_cameras(cameras, scale=1.0):
    """(:94-125) move the world origin to the optical-axis intersection and divide the translations by the LARGEST camera distance (the
    reference overrides its `scale` argument the same way); -1 for a rig whose cameras all sit at the intersection."""
    x, dist, feet, pp, dirs = compute_optical_axis_intersection(cameras)
    far = dist.max()
    if far == 0:
        return -1
    out = cameras.clone()
    out.T = ((x[:, :, None] * out.R).sum(1) + out.T) / far  # world' = world - x  =>  T' = x R + T
    return out, x, feet, pp, dirs


def centerandalign(cameras):
    """(:128-160) one world rotation that turns the rig's mean up vector into +Y; `cameras` a list of single cameras or a batch."""
    rig = join_cameras_as_batch([cameras[i].clone() for i in range(len(cameras))])
    up = rig.R.double()[:, :, 1].mean(0)  # view-space +Y of every camera, in world coordinates
    a = up / up.norm()
    y = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)
    v, c = torch.linalg.cross(a, y), a @ y
    skew = torch.tensor([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]], dtype=torch.float64)
    Q = c * torch.eye(3, dtype=torch.float64) + skew + torch.outer(v, v) / (1.0 + c)  # Q a = y (column vectors)
    # the reference prepends the world rotation `Rotate(Q^T)` to every world-to-view transform: X_view = (X' Q^T) R + T, so R' = Q^T R
    rig.R = torch.einsum("ji,njk->nik", Q.float(), rig.R)
    return rig


def square_bbox(bbox, padding=0.0, astype=None):
    """(:163-185) xyxy box -> the square box around its (rounded) centre whose half side is the larger half extent, padded."""
    kind = type(bbox[0]) if astype is None else astype
    x0, y0, x1, y1 = (np.float64(v) for v in bbox)
    cx, cy = int(np.rint((x0 + x1) / 2)), int(np.rint((y0 + y1) / 2))
    half = int(np.rint(max((x1 - x0) / 2, (y1 - y0) / 2) * (1 + padding)))
    return np.array([cx - half, cy - half, cx + half, cy + half], dtype=kind)


# ---- pytorch3d.implicitron.dataset.utils (third-party, restated; call site data_co3d.py:458-467) ----
def _ndc_to_pixels(focal_length, principal_point, image_size_wh):
    half = image_size_wh / 2
    rescale = half.min()
    return focal_length * rescale, half - principal_point * rescale


def _pixels_to_ndc(focal_length_px, principal_point_px, image_size_wh):
    half = image_size_wh / 2
    rescale = half.min()
    return focal_length_px / rescale, (half - principal_point_px) / rescale


def adjust_camera_to_bbox_crop_(camera, image_size_wh, clamp_bbox_xywh) -> None:
    """In place: intrinsics of a single camera after cropping the image to `clamp_bbox_xywh` (x, y, w, h in pixels)."""
    if len(camera) != 1:
        raise ValueError("Adjusting currently works with singleton cameras camera only")
    image_size_wh, clamp_bbox_xywh = torch.as_tensor(image_size_wh).float(), torch.as_tensor(clamp_bbox_xywh).float()
    f_px, p_px = _ndc_to_pixels(camera.focal_length[0], camera.principal_point[0], image_size_wh)
    f, p = _pixels_to_ndc(f_px, p_px - clamp_bbox_xywh[:2], clamp_bbox_xywh[2:])
    camera.focal_length, camera.principal_point = f[None], p[None]


def adjust_camera_to_image_scale_(camera, original_size_wh, new_size_wh) -> None:
    """In place: intrinsics of a single camera after an aspect-preserving resize `original_size_wh -> new_size_wh`."""
    original_size_wh, out_wh = torch.as_tensor(original_size_wh).float(), torch.as_tensor(new_size_wh).float()
    f_px, p_px = _ndc_to_pixels(camera.focal_length[0], camera.principal_point[0], original_size_wh)
    scale = (out_wh / original_size_wh).min(dim=-1, keepdim=True).values
    f, p = _pixels_to_ndc(f_px * scale, p_px * scale, out_wh)
    camera.focal_length, camera.principal_point = f[None], p[None]


def make_cameras(R, T, focal_lengths, principal_points, original_sizes_wh_wh, crop_coords, image_size: int):
    """The camera construction step of Co3dDataset.__getitem__ (:458-467): one camera per view, intrinsics adjusted for the
    square crop and then for the resize to `image_size`.  `original_sizes_wh_wh` is the loader's `original_size_as_tuple`
    row `[W, H, crop_w, crop_h]`; `crop_coords` its `crop_coords` row (x, y, w, h)."""
    cams = []
    for i in range(len(R)):
        cam = PerspectiveCameras(R=R[i][None], T=T[i][None], focal_length=focal_lengths[i][None], principal_point=principal_points[i][None])
        adjust_camera_to_bbox_crop_(cam, original_sizes_wh_wh[i, :2], crop_coords[i])
        adjust_camera_to_image_scale_(cam, original_sizes_wh_wh[i, 2:], torch.tensor([image_size, image_size]))
        cams.append(cam)
    return cams
