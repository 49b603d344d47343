"""camera conventions either side of the pose path (sgm/data/data_co3d.py camera functions, the sample.py camera
paths of sgm/modules/utils_cameraray.py:317-391, camera.bin).  Golden vectors (tests/golden/cameras.npz) were produced by the
reference's own functions compiled in place (tests/golden/make_golden.py::case_cameras).  The two pytorch3d.implicitron helpers
are third-party and absent, so they are pinned by known answers only.  CPU only."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from cd360 import camera_io, synth
from cd360.cameras import PerspectiveCameras, join_cameras_as_batch, pack_cameras, unpack_cameras
from sgm.data import data_co3d as D
from sgm.modules import utils_cameraray as U

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "cameras.npz")).items()}


def close(a, b, tol=2e-6):
    return torch.allclose(a.float(), b.float(), atol=tol, rtol=1e-5)


def test_normalize_cameras_matches_reference(g):
    rig = unpack_cameras(g["rig"][None])[0]
    new, p_int, p_line, pp, r = D.normalize_cameras(rig)
    assert close(pack_cameras([new])[0], g["norm"]) and close(p_int, g["p_intersect"]) and close(p_line, g["p_line_intersect"])
    assert close(pp, g["pp"]) and close(r, g["r"])
    # what the function is for: optical axes now meet at the origin and the farthest camera is at distance 1
    centres = new.get_camera_center()
    assert abs(float(centres.norm(dim=-1).max()) - 1.0) < 1e-5
    p2, *_ = D.compute_optical_axis_intersection(new)
    assert float(p2.abs().max()) < 1e-5
    # rotations untouched, intrinsics untouched
    assert torch.equal(new.focal_length, rig.focal_length) and close(new.R, rig.R, 1e-7)


def test_normalize_cameras_degenerate_rig_returns_minus_one():
    cam = synth.look_at_camera([0.0, 0.0, -1.0])
    same = join_cameras_as_batch([cam, cam, cam])
    try:
        out = D.normalize_cameras(same)  # identical cameras: every optical axis is the same line
    except AssertionError:
        return  # lstsq of the rank-2 system may produce nan, which the reference asserts on (data_co3d.py:52-54)
    assert out == -1 or isinstance(out, tuple)


def test_centerandalign_matches_reference(g):
    new = unpack_cameras(g["norm"][None])[0]
    aligned = D.centerandalign([new[i] for i in range(len(new))])
    assert close(pack_cameras([aligned])[0], g["aligned"], 5e-6)
    # a global world rotation only: camera-to-origin distances, T and relative rotations are preserved
    assert close(aligned.get_camera_center().norm(dim=-1), new.get_camera_center().norm(dim=-1), 1e-5) and close(aligned.T, new.T, 1e-6)
    rel_a = aligned.R[0].T @ aligned.R[5]
    rel_n = new.R[0].T @ new.R[5]
    assert close(rel_a, rel_n, 1e-5)


def test_square_bbox_matches_reference(g):
    for i, b in enumerate(g["bbox_in"].numpy()):
        assert np.array_equal(D.square_bbox(b, padding=0.1), g["bbox_out"][i].numpy())
        assert np.array_equal(D.square_bbox(b.astype(np.int64), padding=0.0, astype=int), g["bbox_out_int"][i].numpy())


def test_camera_paths_match_reference(g):
    cam1 = unpack_cameras(g["norm"][None])[0][3]
    for axis in "xyz":
        lst = getattr(U, f"interpolate_translate_interpolate_{axis}axis")(cam1, -0.2, 0.21, 0.1)
        got = pack_cameras([join_cameras_as_batch(lst)])[0]
        assert got.shape == g[f"interp_{axis}"].shape and close(got, g[f"interp_{axis}"])
        # the rotation is kept and the centre moved by i along the camera's own axis
        for k, i in enumerate(np.arange(-0.2, 0.21, 0.1)):
            shift_view = lst[k].get_world_to_view_points(cam1.get_camera_center())[0, 0]
            want = torch.zeros(3)
            want["xyz".index(axis)] = -float(i)
            assert close(shift_view, want, 1e-5)
    lf = U.interpolatefocal(cam1, 0.8, 1.25, 0.1)
    assert close(pack_cameras([join_cameras_as_batch(lf)])[0], g["interp_focal"])
    assert len(lf) == len(np.arange(0.8, 1.25, 0.1))


def test_crop_and_scale_intrinsics_known_answers():
    """pytorch3d.implicitron.dataset.utils restated (third-party, absent): known answers.
    * cropping to the full image, or resizing, leaves min-side-normalised NDC intrinsics unchanged for a square image;
    * cropping the centred half-size window doubles the NDC focal length and keeps a centred principal point;
    * a crop shifted by (dx, dy) pixels moves the NDC principal point by (dx, dy)/min(half crop size)."""
    def cam():
        return PerspectiveCameras(R=torch.eye(3)[None], T=torch.zeros(1, 3), focal_length=torch.tensor([[2.0, 2.2]]), principal_point=torch.tensor([[0.1, -0.05]]))
    c = cam()
    D.adjust_camera_to_bbox_crop_(c, torch.tensor([400.0, 400.0]), torch.tensor([0.0, 0.0, 400.0, 400.0]))
    assert close(c.focal_length, cam().focal_length) and close(c.principal_point, cam().principal_point)
    D.adjust_camera_to_image_scale_(c, torch.tensor([400.0, 400.0]), torch.tensor([512, 512]))
    assert close(c.focal_length, cam().focal_length) and close(c.principal_point, cam().principal_point)
    c = PerspectiveCameras(focal_length=torch.tensor([[2.0, 2.0]]), principal_point=torch.zeros(1, 2))
    D.adjust_camera_to_bbox_crop_(c, torch.tensor([400.0, 400.0]), torch.tensor([100.0, 100.0, 200.0, 200.0]))
    assert close(c.focal_length, torch.tensor([[4.0, 4.0]])) and close(c.principal_point, torch.zeros(1, 2))
    c = PerspectiveCameras(focal_length=torch.tensor([[2.0, 2.0]]), principal_point=torch.zeros(1, 2))
    D.adjust_camera_to_bbox_crop_(c, torch.tensor([400.0, 300.0]), torch.tensor([110.0, 40.0, 200.0, 200.0]))
    # principal point in pixels (200,150) -> (90,110) in the crop; NDC = (100-90)/100, (100-110)/100 ; focal 2*150/100
    assert close(c.principal_point, torch.tensor([[0.1, -0.1]])) and close(c.focal_length, torch.tensor([[3.0, 3.0]]))
    with pytest.raises(ValueError):
        D.adjust_camera_to_bbox_crop_(join_cameras_as_batch([cam(), cam()]), torch.tensor([4.0, 4.0]), torch.tensor([0.0, 0, 4, 4]))
    cams = D.make_cameras(torch.eye(3)[None].repeat(2, 1, 1), torch.zeros(2, 3), torch.full((2, 2), 2.0), torch.zeros(2, 2),
                          torch.tensor([[400.0, 300.0, 200.0, 200.0]] * 2), torch.tensor([[110.0, 40.0, 200.0, 200.0]] * 2), 512)
    assert len(cams) == 2 and close(cams[1].principal_point, torch.tensor([[0.1, -0.1]]))


def _fake_pytorch3d_camera_class():
    """A class pickled under the name pytorch3d.renderer.cameras.PerspectiveCameras with the attribute layout pytorch3d's
    TensorProperties(nn.Module) gives its instances (tensor fields as plain attributes beside nn.Module's bookkeeping)."""
    mod = types.ModuleType("pytorch3d.renderer.cameras")

    class PerspectiveCameras(torch.nn.Module):
        def __init__(self, R, T, focal_length, principal_point):
            super().__init__()
            self.device, self._N, self._in_ndc, self.K, self.image_size = torch.device("cpu"), R.shape[0], True, None, None
            self.R, self.T, self.focal_length, self.principal_point = R, T, focal_length, principal_point

    PerspectiveCameras.__module__ = "pytorch3d.renderer.cameras"
    PerspectiveCameras.__qualname__ = "PerspectiveCameras"
    mod.PerspectiveCameras = PerspectiveCameras
    return mod


def test_camera_bin_written_by_the_reference_layout_loads_without_pytorch3d(tmp_path):
    ring = synth.ring_cameras(6, seed=5)
    mod = _fake_pytorch3d_camera_class()
    pkgs = {"pytorch3d": types.ModuleType("pytorch3d"), "pytorch3d.renderer": types.ModuleType("pytorch3d.renderer"), "pytorch3d.renderer.cameras": mod}
    saved = {k: sys.modules.get(k) for k in pkgs}
    sys.modules.update(pkgs)
    try:
        mk = lambda c: mod.PerspectiveCameras(c.R, c.T, c.focal_length, c.principal_point)
        torch.save([[mk(ring[i]) for i in range(2)], [mk(ring[i]) for i in range(2, 6)]], tmp_path / "camera.bin")  # main.py:1025-1029
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert "pytorch3d" not in sys.modules
    val, train = camera_io.load_camera_bin(tmp_path / "camera.bin")
    assert len(val) == 2 and len(train) == 4 and all(isinstance(c, PerspectiveCameras) for c in val + train)
    assert torch.equal(pack_cameras([join_cameras_as_batch(val + train)])[0], pack_cameras([join_cameras_as_batch(ring)])[0])
    # own writer round-trips too
    camera_io.save_camera_bin(tmp_path / "own.bin", val, train)
    val2, train2 = camera_io.load_camera_bin(tmp_path / "own.bin")
    assert torch.equal(train2[3].T, train[3].T) and len(val2) == 2
    with pytest.raises(ValueError):
        torch.save([1, 2, 3], tmp_path / "bad.bin")
        camera_io.load_camera_bin(tmp_path / "bad.bin")


def test_reference_choices_and_pose_batches_follow_sample_py():
    assert camera_io.reference_view_choices(50, 8) == [int(x) for x in torch.linspace(0, 50 - 50 / 8, 8)]
    assert camera_io.reference_view_choices(50, 50) == list(range(50))
    train = synth.ring_cameras(20, seed=1)
    val = synth.ring_cameras(3, seed=2, phase=0.3)
    choices = camera_io.reference_view_choices(20, 4)
    batches = camera_io.sampling_pose_batches([val[0], val[2]], train, choices)
    assert len(batches) == 2 and len(batches[0]["pose"]) == 1 and len(batches[0]["pose"][0]) == 5
    assert torch.equal(batches[1]["pose"][0].T[0], val[2].T[0]) and torch.equal(batches[1]["pose"][0].T[2], train[choices[1]].T[0])
    assert batches[0]["original_size_as_tuple"].tolist() == [[512, 512]]
    moved = camera_io.sampling_pose_batches([val[0]], train, choices, path="y")  # sample.py defaults: arange(-0.2, 0.21, 0.4)
    assert len(moved) == len(np.arange(-0.2, 0.21, 0.4)) == 2
    assert not torch.equal(moved[0]["pose"][0].T[0], moved[1]["pose"][0].T[0]) and torch.equal(moved[0]["pose"][0].T[1:], moved[1]["pose"][0].T[1:])
    packed = pack_cameras(moved[0]["pose"])
    assert packed.shape == (1, 5, 16)


def test_camera_methods_against_an_independent_homogeneous_matrix_restatement():
    """pytorch3d is not in the reference tree, and the golden generator stands `cd360.cameras` in for it (tests/golden/refshim.py): a
    convention error there would be shared by goldens, oracle and kernels.  Second opinion, written independently of cd360/cameras.py's
    ordered fp32 chains: PerspectiveCameras as pytorch3d PUBLISHES it -- row-vector 4 x 4 world-to-view transform [[R, 0], [T, 1]], NDC
    calibration matrix K = [[fx, 0, px, 0], [0, fy, py, 0], [0, 0, 0, 1], [0, 0, 1, 0]] applied as [x y z 1] K^T followed by the division
    by the last coordinate, unprojection as the inverse of that composite, camera centre as the world point mapped to the view origin -- in
    float64 homogeneous matrices, against every camera method the path uses, on random rigs."""
    g = torch.Generator().manual_seed(11)
    rig = join_cameras_as_batch(synth.ring_cameras(7, seed=13))
    n = len(rig)
    R, T = rig.R.double(), rig.T.double()
    f, pp = rig.focal_length.double(), (rig.principal_point + 0.05 * torch.randn(n, 2, generator=g)).double()
    rig.principal_point = pp.float()
    w2v = torch.zeros(n, 4, 4, dtype=torch.float64)
    w2v[:, :3, :3], w2v[:, 3, :3], w2v[:, 3, 3] = R, T, 1.0
    K = torch.zeros(n, 4, 4, dtype=torch.float64)
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 2, 3], K[:, 3, 2] = f[:, 0], f[:, 1], pp[:, 0], pp[:, 1], 1.0, 1.0
    pts = torch.randn(5, 3, generator=g).double() * 0.3
    hom = torch.cat([pts, torch.ones(5, 1, dtype=torch.float64)], 1)           # [P, 4]
    view = torch.einsum("pi,nij->npj", hom, w2v)                                # [n, P, 4] row vectors
    clip = torch.einsum("npi,nji->npj", view, K)                                # x K^T
    ndc = clip[..., :3] / clip[..., 3:]                                         # (fx X/Z + px, fy Y/Z + py, 1/Z)
    assert close(rig.get_world_to_view_points(pts.float()), view[..., :3], 1e-5)
    assert close(rig.transform_points_ndc(pts.float()), ndc, 2e-5)
    # unprojection of (x_ndc, y_ndc, depth): invert the composite on (x z, y z, 1, z) -- the clip vector of that NDC point at depth z
    xy_depth = torch.cat([ndc[..., :2], view[..., 2:3]], -1)                    # what project() produced, depth = Z_view
    z = xy_depth[..., 2:3]
    clip_back = torch.cat([xy_depth[..., :2] * z, torch.ones_like(z), z], -1)
    world = torch.einsum("npi,nij->npj", clip_back, torch.linalg.inv(torch.einsum("nij,nkj->nik", w2v, K)))
    for i in range(n):
        got = rig[i].unproject_points(xy_depth[i].float(), world_coordinates=True)
        assert close(got.reshape(-1, 3), pts, 2e-5) and close(world[i, :, :3] / world[i, :, 3:], pts, 1e-9)
    centre = torch.linalg.inv(w2v)[:, 3, :3]                                     # the world point with view coordinates (0, 0, 0)
    assert close(rig.get_camera_center(), centre, 1e-5)
    # axis conventions (+X left, +Y up, +Z into the scene): a point to the camera's own +X lands at POSITIVE ndc x
    eye = PerspectiveCameras(R=torch.eye(3)[None], T=torch.zeros(1, 3), focal_length=torch.tensor([[2.0, 2.0]]), principal_point=torch.zeros(1, 2))
    assert float(eye.transform_points_ndc(torch.tensor([[0.1, 0.0, 1.0]]))[0, 0, 0]) > 0
    assert float(eye.transform_points_ndc(torch.tensor([[0.0, 0.1, 1.0]]))[0, 0, 1]) > 0
