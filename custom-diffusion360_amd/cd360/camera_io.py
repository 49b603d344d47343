"""`camera.bin` and the pose batches of sample.py, without pytorch3d.

The reference writes `torch.save([cameras_val, cameras_train], f'{logdir}/camera.bin')` (main.py:1025-1029) where both entries
are LISTS of single-view pytorch3d `PerspectiveCameras` objects, and reads it back with `torch.load` (sample.py:273).  The
pickle therefore names `pytorch3d.renderer.cameras.PerspectiveCameras`; this module loads such a file on a box with no pytorch3d
by resolving that name (and any other `pytorch3d.*` class) to a state-capturing shim and lifting `R, T, focal_length,
principal_point` out of the pickled attribute dictionary (pytorch3d's TensorProperties keeps them as plain tensor attributes).

`reference_view_choices` / `sampling_pose_batches` restate sample.py:274-326 — which training views become the `n` references
and how target + references are joined into one camera batch per image (the `pose` list the UNet receives).
"""
from __future__ import annotations

import copy
import pickle
from typing import List, Sequence

import torch

from .cameras import PerspectiveCameras, join_cameras_as_batch

_FIELDS = ("R", "T", "focal_length", "principal_point")


class _PickledObject:
    """Whatever class a pickle names under pytorch3d.*: keeps the instance dictionary so the camera fields can be read."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


def _as_camera(obj) -> PerspectiveCameras:
    if isinstance(obj, PerspectiveCameras):
        return obj
    d = getattr(obj, "__dict__", {})
    missing = [k for k in ("R", "T") if k not in d]
    if missing:
        raise ValueError(f"pickled camera object has no {missing} (keys: {sorted(d)[:12]})")
    focal = d.get("focal_length", 1.0)
    pp = d.get("principal_point", ((0.0, 0.0),))
    if d.get("_in_ndc", True) is False:
        raise NotImplementedError("screen-space (in_ndc=False) cameras are not used by the reference and are not restated")
    return PerspectiveCameras(R=d["R"], T=d["T"], focal_length=focal, principal_point=pp)


class _ShimPickle:
    """`pickle_module` for torch.load: the stdlib pickle with pytorch3d.* classes resolved to _PickledObject."""

    __name__ = "cd360_camera_pickle"
    Unpickler = None  # set below
    load = staticmethod(pickle.load)
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "pytorch3d" or module.startswith("pytorch3d."):
            return type(name, (_PickledObject,), {"__module__": module})
        if module == "cd360.cameras" and name == "PerspectiveCameras":
            return PerspectiveCameras
        return super().find_class(module, name)


_ShimPickle.Unpickler = _Unpickler


def load_camera_bin(path) -> tuple[List[PerspectiveCameras], List[PerspectiveCameras]]:
    """-> (cameras_val, cameras_train): two lists of single-view cameras, as sample.py:273 unpacks them."""
    obj = torch.load(path, map_location="cpu", pickle_module=_ShimPickle, weights_only=False)
    if not (isinstance(obj, (list, tuple)) and len(obj) == 2):
        raise ValueError("camera.bin must hold [cameras_val, cameras_train]")

    def as_list(x):
        if isinstance(x, (list, tuple)):
            return [_as_camera(c) for c in x]
        cam = _as_camera(x)  # a single batched object: split it per view
        return [cam[i] for i in range(len(cam))]

    return as_list(obj[0]), as_list(obj[1])


def save_camera_bin(path, cameras_val: Sequence[PerspectiveCameras], cameras_train: Sequence[PerspectiveCameras]) -> None:
    """Writes the same two-list structure with this package's pytorch3d-free camera class (readable by load_camera_bin)."""
    torch.save([list(cameras_val), list(cameras_train)], path)


def reference_view_choices(num_train: int, num_ref: int = 8) -> List[int]:
    """sample.py:275-277: `num_ref` training views evenly spread over the training ring (the `choices` that also index the
    per-block `references` buffer, sample.py:91)."""
    max_diff = num_train / num_ref
    return [int(x) for x in torch.linspace(0, num_train - max_diff, num_ref)]


def sampling_pose_batches(targets: Sequence[PerspectiveCameras], cameras_train: Sequence[PerspectiveCameras], choices: Sequence[int],
                          path: str | None = None, interp_start: float = -0.2, interp_end: float = 0.21, interp_step: float = 0.4):
    """sample.py:299-326: one batch dict per generated image; `pose` = [join(target, *references)].  `path` in
    {None, 'x', 'y', 'z', 'focal'} selects the camera path applied to each target (translateX/Y/Z, translate_focal)."""
    from sgm.modules import utils_cameraray as ucr

    refs = [cameras_train[i] for i in choices]
    fn = {None: None, "x": ucr.interpolate_translate_interpolate_xaxis, "y": ucr.interpolate_translate_interpolate_yaxis,
          "z": ucr.interpolate_translate_interpolate_zaxis, "focal": ucr.interpolatefocal}[path]
    size = lambda: torch.tensor([512, 512]).reshape(-1, 2)
    zero = lambda: torch.tensor([0, 0]).reshape(-1, 2)
    batches = []
    for tgt in targets:
        base = {"original_size_as_tuple": size(), "target_size_as_tuple": size(), "crop_coords_top_left": zero(),
                "original_size_as_tuple_ref": size(), "target_size_as_tuple_ref": size(), "crop_coords_top_left_ref": zero()}
        for cam in ([tgt] if fn is None else fn(tgt, interp_start, interp_end, interp_step)):
            b = copy.deepcopy(base)
            b["pose"] = [join_cameras_as_batch([cam] + refs)]
            batches.append(b)
    return batches
