"""Synthetic "car0-shaped" camera rings and inputs.

There are no CO3D assets or checkpoints in this environment, so benchmarks and tests use
cameras on a ring around the origin with the conventions `normalize_cameras`
(reference sgm/data/data_co3d.py:94-125) leaves behind: farthest camera at radius 1,
optical axes through the origin, NDC focal length ~2.
"""
from __future__ import annotations

import math

import torch

from .cameras import PerspectiveCameras, join_cameras_as_batch


def look_at_camera(center, at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), focal=2.0, pp=(0.0, 0.0)) -> PerspectiveCameras:
    """Camera at `center` looking at `at` in the PyTorch3D convention (X_view = X_world @ R + T)."""
    c = torch.tensor(center, dtype=torch.float64)
    a = torch.tensor(at, dtype=torch.float64)
    u = torch.tensor(up, dtype=torch.float64)
    z = (a - c) / (a - c).norm()
    x = torch.linalg.cross(u, z)
    x = x / x.norm()
    y = torch.linalg.cross(z, x)
    R = torch.stack([x, y, z], dim=1)  # columns = camera axes in world coordinates
    T = -(c @ R)
    return PerspectiveCameras(focal_length=[[focal, focal]], principal_point=[list(pp)], R=R.float()[None], T=T.float()[None])


def ring_cameras(num: int, seed: int = 0, radius=(0.8, 1.0), elev_deg=(5.0, 25.0), focal=(2.0, 2.4), phase: float = 0.0):
    """`num` cameras on a ring, deterministic for a given seed. Returns a list of batch-1 cameras."""
    g = torch.Generator().manual_seed(seed)
    cams = []
    for i in range(num):
        az = 2.0 * math.pi * (i + phase) / num
        rad = radius[0] + (radius[1] - radius[0]) * torch.rand(1, generator=g).item()
        el = math.radians(elev_deg[0] + (elev_deg[1] - elev_deg[0]) * torch.rand(1, generator=g).item())
        fl = focal[0] + (focal[1] - focal[0]) * torch.rand(1, generator=g).item()
        ppx, ppy = (0.02 * (torch.rand(2, generator=g) - 0.5)).tolist()
        center = (rad * math.cos(el) * math.sin(az), rad * math.sin(el), rad * math.cos(el) * math.cos(az))
        cams.append(look_at_camera(center, focal=fl, pp=(ppx, ppy)))
    return cams


def pose_batch(b: int, n_ref: int, seed: int = 0, n_train: int | None = None):
    """`pose` exactly as the reference passes it: a list (len b) of camera batches of size n_ref+1,
    element 0 = target view, 1.. = reference views (sample.py:302,326)."""
    n_train = n_train or max(n_ref, 8)
    train = ring_cameras(n_train, seed=seed)
    val = ring_cameras(max(b, 4), seed=seed + 1, phase=0.37)
    choices = [int(x) for x in torch.linspace(0, n_train - n_train / n_ref, n_ref)]
    refs = [train[i] for i in choices]
    return [join_cameras_as_batch([val[i % len(val)]] + refs) for i in range(b)]
