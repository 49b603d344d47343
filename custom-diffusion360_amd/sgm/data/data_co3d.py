"""Camera normalisation and crop/scale intrinsics of the reference's CO3D loader, pytorch3d-free.

Mirrors sgm/data/data_co3d.py:27-160 (intersect_skew_lines_high_dim, compute_optical_axis_intersection, normalize_cameras,
centerandalign, square_bbox) and the two pytorch3d.implicitron.dataset.utils helpers the loader calls at :458-467
(adjust_camera_to_bbox_crop_, adjust_camera_to_image_scale_).  pytorch3d is an un-pinned third-party dependency that is not in
the reference tree; its NDC<->pixel conversion is restated from the published algorithm (min-side-normalised NDC:
p_px = half_size - p_ndc * min(half_size), f_px = f_ndc * min(half_size)) and is pinned here only by known-answer tests.

These run once per dataset on a few hundred cameras: host fp32 torch on CPU, exactly like the reference.  They produce the
`[R, T, focal, principal point]` rows that cd360.cameras.pack_cameras hands to the HIP path.
"""
from __future__ import annotations

import numpy as np
import torch

from cd360.cameras import PerspectiveCameras, Rotate, Translate, join_cameras_as_batch


def intersect_skew_lines_high_dim(p, r, mask=None):
    """Least-squares point closest to a bundle of lines (p + t r): solve [sum (I - r r^T)] x = sum (I - r r^T) p (:39-55)."""
    dim = p.shape[-1]
    if mask is None:
        mask = torch.ones_like(p[..., 0])
    r = torch.nn.functional.normalize(r, dim=-1)
    eye = torch.eye(dim, device=p.device, dtype=p.dtype)[None, None]
    i_min_cov = (eye - (r[..., None] * r[..., None, :])) * mask[..., None, None]
    sum_proj = i_min_cov.matmul(p[..., None]).sum(dim=-3)
    p_intersect = torch.linalg.lstsq(i_min_cov.sum(dim=-3), sum_proj).solution[..., 0]
    if torch.any(torch.isnan(p_intersect)):
        raise AssertionError(f"degenerate camera bundle: {p_intersect}")
    return p_intersect, r


def _point_line_distance(p1, r1, p2):
    df = p2 - p1
    proj_vector = df - ((df * r1).sum(dim=-1, keepdim=True) * r1)
    return proj_vector.norm(dim=-1), p2 - proj_vector


def intersect_skew_line_groups(p, r, mask):
    """(:27-36) intersection point, its foot on every line, squared distances, normalised directions."""
    p_intersect, r = intersect_skew_lines_high_dim(p, r, mask=mask)
    _, p_line_intersect = _point_line_distance(p, r, p_intersect[..., None, :].expand_as(p))
    dist2 = ((p_line_intersect - p_intersect[..., None, :]) ** 2).sum(dim=-1)
    return p_intersect, p_line_intersect, dist2, r


def compute_optical_axis_intersection(cameras):
    """(:66-91) optical axis of camera i = line from its centre through the unprojection of (principal point, depth 1)."""
    centers = cameras.get_camera_center()
    n = len(cameras)
    axis_ndc = torch.cat((cameras.principal_point, torch.ones((n, 1))), -1)
    pp = cameras.unproject_points(axis_ndc, from_ndc=True, world_coordinates=True)  # [n cameras, n points, 3]
    pp2 = pp[torch.arange(n), torch.arange(n)]  # camera i applied to its own point i
    directions = (pp2 - centers)[None, None]
    centers = centers[None, None]
    p_intersect, p_line_intersect, _, r = intersect_skew_line_groups(p=centers, r=directions, mask=None)
    p_intersect = p_intersect.squeeze().unsqueeze(0)
    dist = (p_intersect - centers).norm(dim=-1)
    return p_intersect, dist, p_line_intersect, pp2, r


def normalize_cameras(cameras, scale=1.0):
    """(:94-125) move the world origin to the optical-axis intersection and divide translations by the LARGEST camera
    distance (the reference overrides its `scale` argument with max(dist)); returns -1 for a degenerate (zero-scale) rig."""
    new_cameras = cameras.clone()
    new_transform = new_cameras.get_world_to_view_transform()
    p_intersect, dist, p_line_intersect, pp, r = compute_optical_axis_intersection(cameras)
    t = Translate(p_intersect)
    scale = max(dist.squeeze())
    if scale == 0:
        return -1
    new_transform = t.compose(new_transform)
    new_cameras.R = new_transform.get_matrix()[:, :3, :3]
    new_cameras.T = new_transform.get_matrix()[:, 3, :3] / scale
    return new_cameras, p_intersect, p_line_intersect, pp, r


def centerandalign(cameras):
    """(:128-160) rotate the world so the rig's mean up vector becomes +Y (Rodrigues form of the a->b alignment)."""
    new_cameras = join_cameras_as_batch([cameras[i].clone() for i in range(len(cameras))])
    cam_trans = new_cameras.get_world_to_view_transform().inverse()
    eye_at_up_view = torch.tensor([[0, 0, 0], [0, 0, 1], [0, 1, 0]], dtype=torch.float32)
    eye_at_up_world = cam_trans.transform_points(eye_at_up_view).reshape(-1, 3, 3)
    eye, _at, up_plus_eye = eye_at_up_world.unbind(1)
    up = torch.mean(up_plus_eye - eye, dim=0).numpy()
    n = up / np.linalg.norm(up)
    v = np.cross(n, [0, 1, 0])
    s = np.linalg.norm(v)
    c = np.dot(n, [0, 1, 0])
    V = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    rot = torch.from_numpy(np.eye(3) + V + V @ V * (1 - c) / s**2).float()
    new_transform = Rotate(rot.T).compose(new_cameras.get_world_to_view_transform())
    new_cameras.R = new_transform.get_matrix()[:, :3, :3]
    new_cameras.T = new_transform.get_matrix()[:, 3, :3]
    return new_cameras


def square_bbox(bbox, padding=0.0, astype=None):
    """(:163-185) xyxy box -> centred square box with optional relative padding."""
    if astype is None:
        astype = type(bbox[0])
    bbox = np.array(bbox)
    center = ((bbox[:2] + bbox[2:]) / 2).round().astype(int)
    extents = (bbox[2:] - bbox[:2]) / 2
    s = (max(extents) * (1 + padding)).round().astype(int)
    return np.array([center[0] - s, center[1] - s, center[0] + s, center[1] + s], dtype=astype)


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
