"""pytorch3d-free pin-hole camera container for the pose path.

The reference hands `pose` around as lists of `pytorch3d.renderer.cameras.PerspectiveCameras`
batches (sample.py:302-326, sgm/modules/nerfsd_pytorch3d.py:73-77,
sgm/modules/utils_cameraray.py:61-196).  pytorch3d is a third-party dependency that is
not in the reference tree, so its conventions are restated here from SURVEY.md Appendix B:

  * row-vector convention        X_view = X_world @ R + T
  * NDC, +X left, +Y up, +Z in   x_ndc = fx * X/Z + px ;  y_ndc = fy * Y/Z + py
  * unproject(x, y, depth)       X_view = ((x-px)*depth/fx, (y-py)*depth/fy, depth)
                                 X_world = (X_view - T) @ R^T
  * camera centre                C = -T @ R^T

Every formula below is written as an explicit, ordered chain of fp32 multiplies and adds
(no matmul) so that the oracle (oracle/pose_path.py) and the HIP kernels
(csrc/ray_project.hip, compiled with -ffp-contract=off) can reproduce it bit for bit; the
integer bilinear corner indices downstream depend on it.

The HIP path never touches these Python methods: it consumes `pack_cameras(pose)`, a
`[b, n+1, 16]` fp32 tensor (R row-major 9, T 3, focal 2, principal point 2).
"""
from __future__ import annotations

from typing import List, Sequence

import torch

PACKED_CAMERA_FLOATS = 16


class PerspectiveCameras:
    """Minimal stand-in for pytorch3d's PerspectiveCameras (NDC convention only)."""

    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), R=None, T=None, device="cpu", **_ignored):
        def as2d(v, width):
            t = torch.as_tensor(v, dtype=torch.float32)
            if t.dim() == 0:
                t = t.reshape(1, 1).expand(1, width)
            elif t.dim() == 1:
                t = t.reshape(1, -1) if t.numel() == width else t.reshape(-1, 1).expand(-1, width)
            return t.clone()

        if R is None:
            R = torch.eye(3, dtype=torch.float32)[None]
        if T is None:
            T = torch.zeros(1, 3, dtype=torch.float32)
        R = torch.as_tensor(R, dtype=torch.float32)
        T = torch.as_tensor(T, dtype=torch.float32)
        if R.dim() == 2:
            R = R[None]
        if T.dim() == 1:
            T = T[None]
        n = max(R.shape[0], T.shape[0])
        f = as2d(focal_length, 2)
        p = as2d(principal_point, 2)
        self.R = R.expand(n, 3, 3).clone().to(device)
        self.T = T.expand(n, 3).clone().to(device)
        self.focal_length = f.expand(n, 2).clone().to(device)
        self.principal_point = p.expand(n, 2).clone().to(device)

    # ---- container protocol used by the reference (len(), cameras[i], .to(), .device) ----
    def __len__(self) -> int:
        return self.R.shape[0]

    @property
    def device(self):
        return self.R.device

    def __getitem__(self, idx) -> "PerspectiveCameras":
        if isinstance(idx, int):
            idx = [idx]
        return PerspectiveCameras(
            focal_length=self.focal_length[idx],
            principal_point=self.principal_point[idx],
            R=self.R[idx],
            T=self.T[idx],
            device=self.device,
        )

    def to(self, device) -> "PerspectiveCameras":
        return PerspectiveCameras(
            focal_length=self.focal_length, principal_point=self.principal_point, R=self.R, T=self.T, device=device
        )

    def clone(self) -> "PerspectiveCameras":
        return self.to(self.device)

    # ---- geometry (ordered fp32 chains; see module docstring) ----
    def get_camera_center(self) -> torch.Tensor:
        R, T = self.R, self.T
        n0, n1, n2 = -T[:, 0:1], -T[:, 1:2], -T[:, 2:3]
        return (n0 * R[:, :, 0] + n1 * R[:, :, 1]) + n2 * R[:, :, 2]

    def unproject_points(self, xy_depth, world_coordinates=True, from_ndc=True, **_ignored) -> torch.Tensor:
        assert from_ndc, "only the NDC convention is restated"
        p = xy_depth.to(self.device).reshape(1, -1, 3) if xy_depth.dim() == 2 else xy_depth.to(self.device)
        f, c = self.focal_length[:, None, :], self.principal_point[:, None, :]
        d = p[..., 2:3]
        xv = ((p[..., 0:1] - c[..., 0:1]) * d) / f[..., 0:1]
        yv = ((p[..., 1:2] - c[..., 1:2]) * d) / f[..., 1:2]
        zv = d.expand_as(xv)
        if not world_coordinates:
            return torch.cat([xv, yv, zv], -1)
        R, T = self.R[:, None], self.T[:, None]
        a0, a1, a2 = xv - T[..., 0:1], yv - T[..., 1:2], zv - T[..., 2:3]
        return (a0 * R[..., :, 0] + a1 * R[..., :, 1]) + a2 * R[..., :, 2]

    def get_world_to_view_points(self, points) -> torch.Tensor:
        p = points.to(self.device)
        if p.dim() == 2:
            p = p[None]
        R, T = self.R[:, None], self.T[:, None]
        return ((p[..., 0:1] * R[..., 0, :] + p[..., 1:2] * R[..., 1, :]) + p[..., 2:3] * R[..., 2, :]) + T

    def get_world_to_view_transform(self) -> "Transform3d":
        """pytorch3d's row-vector 4x4: [[R, 0], [T, 1]] so that [X_world, 1] @ M = [X_view, 1]."""
        n = len(self)
        m = torch.zeros(n, 4, 4, dtype=torch.float32, device=self.device)
        m[:, :3, :3], m[:, 3, :3], m[:, 3, 3] = self.R, self.T, 1.0
        return Transform3d(m)

    def transform_points_ndc(self, points, **_ignored) -> torch.Tensor:
        v = self.get_world_to_view_points(points)
        f, c = self.focal_length[:, None, :], self.principal_point[:, None, :]
        x = (f[..., 0:1] * v[..., 0:1]) / v[..., 2:3] + c[..., 0:1]
        y = (f[..., 1:2] * v[..., 1:2]) / v[..., 2:3] + c[..., 1:2]
        return torch.cat([x, y, 1.0 / v[..., 2:3]], -1)


class Transform3d:
    """The slice of pytorch3d.transforms.Transform3d the reference's camera utilities use (data_co3d.py:94-160,
    utils_cameraray.py:317-375): row-vector 4x4 matrices, `a.compose(b)` = apply a then b = a.M @ b.M."""

    def __init__(self, matrix=None, device="cpu"):
        self._m = torch.eye(4, dtype=torch.float32, device=device)[None] if matrix is None else matrix.to(torch.float32)

    def get_matrix(self) -> torch.Tensor:
        return self._m

    @property
    def device(self):
        return self._m.device

    def compose(self, *others: "Transform3d") -> "Transform3d":
        m = self._m
        for o in others:
            m = m @ o.get_matrix()
        return Transform3d(m)

    def inverse(self) -> "Transform3d":
        return Transform3d(torch.linalg.inv(self._m))

    def transform_points(self, points: torch.Tensor) -> torch.Tensor:
        p = points.to(self._m.device, torch.float32)
        squeeze = p.dim() == 2
        if squeeze:
            p = p[None]
        h = torch.cat([p, torch.ones_like(p[..., :1])], -1) @ self._m
        out = h[..., :3] / h[..., 3:]
        return out[0] if squeeze and out.shape[0] == 1 else out  # pytorch3d: (P,3) in, one transform -> (P,3) out

    def __len__(self):
        return self._m.shape[0]


def Translate(xyz: torch.Tensor) -> Transform3d:
    xyz = torch.as_tensor(xyz, dtype=torch.float32).reshape(-1, 3)
    m = torch.eye(4, dtype=torch.float32, device=xyz.device)[None].repeat(xyz.shape[0], 1, 1)
    m[:, 3, :3] = xyz
    return Transform3d(m)


def Rotate(R: torch.Tensor) -> Transform3d:
    R = torch.as_tensor(R, dtype=torch.float32)
    R = R[None] if R.dim() == 2 else R
    m = torch.eye(4, dtype=torch.float32, device=R.device)[None].repeat(R.shape[0], 1, 1)
    m[:, :3, :3] = R
    return Transform3d(m)


def join_cameras_as_batch(cameras: Sequence[PerspectiveCameras]) -> PerspectiveCameras:
    return PerspectiveCameras(
        focal_length=torch.cat([c.focal_length for c in cameras]),
        principal_point=torch.cat([c.principal_point for c in cameras]),
        R=torch.cat([c.R for c in cameras]),
        T=torch.cat([c.T for c in cameras]),
        device=cameras[0].device,
    )


def pack_cameras(pose, device=None) -> torch.Tensor:
    """`pose` (list of length b of camera batches of size n+1, or an already packed tensor)
    -> contiguous fp32 `[b, n+1, 16]` = (R row-major 9 | T 3 | focal 2 | principal point 2).

    Accepts anything exposing `.R [m,3,3] .T [m,3] .focal_length [m,2] .principal_point [m,2]`
    (this module's class or a real pytorch3d PerspectiveCameras)."""
    if isinstance(pose, torch.Tensor):
        assert pose.shape[-1] == PACKED_CAMERA_FLOATS
        out = pose.to(torch.float32)
    else:
        rows = []
        for cams in pose:
            m = cams.R.shape[0]
            f = torch.as_tensor(cams.focal_length, dtype=torch.float32).reshape(-1, cams.focal_length.shape[-1] if cams.focal_length.dim() > 1 else 1)
            if f.shape[-1] == 1:
                f = f.expand(-1, 2)
            rows.append(
                torch.cat(
                    [
                        cams.R.reshape(m, 9).float().cpu(),
                        cams.T.reshape(m, 3).float().cpu(),
                        f.expand(m, 2).float().cpu(),
                        cams.principal_point.reshape(-1, 2).expand(m, 2).float().cpu(),
                    ],
                    dim=1,
                )
            )
        out = torch.stack(rows)
    if device is not None:
        out = out.to(device)
    return out.contiguous()


def unpack_cameras(packed: torch.Tensor) -> List[PerspectiveCameras]:
    """Inverse of `pack_cameras` (CPU objects)."""
    packed = packed.detach().float().cpu()
    return [
        PerspectiveCameras(
            focal_length=row[:, 12:14], principal_point=row[:, 14:16], R=row[:, :9].reshape(-1, 3, 3), T=row[:, 9:12]
        )
        for row in packed
    ]
