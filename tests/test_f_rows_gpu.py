"""SURVEY.md section 8 rows f1 and f2 closed on the GPU: the callers either side of the hot path carried INTO the HIP kernels and
compared with the oracle / the reference's goldens (the CPU halves live in tests/test_cameras_cpu.py and tests/test_sampler_cpu.py).
Needs an MI355X."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import weights as W
from cd360.cameras import join_cameras_as_batch, pack_cameras, unpack_cameras

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


def _reference_layout_camera_bin(path, cams_val, cams_train):
    """main.py:1025-1029: torch.save([cameras_val, cameras_train]) of lists of single-view objects pickled under the name
    pytorch3d.renderer.cameras.PerspectiveCameras, with the attribute layout pytorch3d's TensorProperties(nn.Module) gives them."""
    mod = types.ModuleType("pytorch3d.renderer.cameras")

    class PerspectiveCameras(torch.nn.Module):
        def __init__(self, c):
            super().__init__()
            self.device, self._N, self._in_ndc, self.K, self.image_size = torch.device("cpu"), c.R.shape[0], True, None, None
            self.R, self.T, self.focal_length, self.principal_point = c.R, c.T, c.focal_length, c.principal_point

    PerspectiveCameras.__module__, PerspectiveCameras.__qualname__ = "pytorch3d.renderer.cameras", "PerspectiveCameras"
    mod.PerspectiveCameras = PerspectiveCameras
    pkgs = {"pytorch3d": types.ModuleType("pytorch3d"), "pytorch3d.renderer": types.ModuleType("pytorch3d.renderer"), "pytorch3d.renderer.cameras": mod}
    saved = {k: sys.modules.get(k) for k in pkgs}
    sys.modules.update(pkgs)
    try:
        torch.save([[PerspectiveCameras(c) for c in cams_val], [PerspectiveCameras(c) for c in cams_train]], path)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@torch.no_grad()
def test_f1_normalised_rig_through_camera_bin_into_the_hip_render(tmp_path):
    """f1 end to end: the off-centre, off-scale 12-camera rig of tests/golden/cameras.npz -> data_co3d.normalize_cameras (pinned on the
    reference's own output in the same fixture) -> a `camera.bin` in the REFERENCE's pickle layout (pytorch3d class names; loaded here
    without pytorch3d) -> camera_io.reference_view_choices / sampling_pose_batches (sample.py:274-326) -> the `pose` list of one image
    -> ONE pose block on the HIP kernels (FeatureNeRF render at the target camera, pose-token attention, volume render, injection).
    Against the oracle on the SAME packed cameras: block output / fg / alphas / rgb within the bf16 bar, and the integer bilinear corner
    indices, in-bounds masks, grid coordinates, rays and sample points of every (view, ray, sample) BIT-EXACT."""
    from cd360 import camera_io, nerf, ops
    from oracle import pose_path as O
    from sgm.data import data_co3d as D
    from sgm.modules.attention import BasicTransformerBlock
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "cameras.npz")).items()}
    rig = unpack_cameras(g["rig"][None])[0]
    new = D.normalize_cameras(rig)[0]
    assert torch.allclose(pack_cameras([new])[0], g["norm"], atol=2e-6, rtol=1e-5)  # the reference's normalize_cameras output
    singles = [new[i] for i in range(len(new))]
    _reference_layout_camera_bin(tmp_path / "camera.bin", singles[:2], singles[2:])
    assert "pytorch3d" not in sys.modules
    val, train = camera_io.load_camera_bin(tmp_path / "camera.bin")
    assert torch.equal(pack_cameras([join_cameras_as_batch(val + train)])[0], pack_cameras([new])[0])  # the file round trip is exact
    n = 4
    choices = camera_io.reference_view_choices(len(train), n)
    batches = camera_io.sampling_pose_batches([val[1]], train, choices, path="x", interp_start=-0.1, interp_end=0.11, interp_step=0.2)
    assert len(batches) == 2  # two target cameras on the translate-x path of sample.py:305-309
    C, heads, cd, S, r = 64, 1, 32, 4, 8
    hw = r * r
    blk = BasicTransformerBlock(C, heads, 64, context_dim=cd, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=S, rgb_predict=True, mode="feature-nerf", stratified=True).eval()
    w = {k: v.to(BF).float() for k, v in W.load_into(blk, seed=17).items()}
    blk = blk.to(DEV, BF)
    x = W.tensor("f1.x", (1, hw, C), seed=17).to(BF)
    ctx = W.tensor("f1.ctx", (1, 77, cd), seed=17).to(BF)
    cref = W.tensor("f1.cref", (1, n, hw, C), seed=17).to(BF)
    for bi, batch in enumerate(batches):
        pose = batch["pose"]
        cams = pack_cameras(pose)  # [1, n + 1, 16] fp32: what the kernels read
        out, fg, _, alphas, rgb = blk(x.to(DEV), context=ctx.to(DEV), context_ref=cref.reshape(n, hw, C).to(DEV), pose=pose)
        want = O.transformer_block(w, x.float(), ctx.float(), heads, context_ref=cref.float(), cams=cams, num_samples=S, far=2.0)
        errs = (rel(out, want[0]), rel(fg, want[1]), rel(alphas, want[2]), rel(rgb, want[3]))
        print(f"f1 pose {bi}: block out / fg / alphas / rgb vs oracle:", [round(e, 5) for e in errs])
        assert max(errs) < 1e-2, errs
        # the integer side: bit-exact
        xs = nerf.patch_positions(r, DEV)
        t, _ = nerf.depth_samples(S, 2.0, 0.0, DEV, hw)
        res = ops.ray_project_index(cams.to(DEV), xs, xs, t)
        xo = O.patch_positions(r)
        rays = O.patch_rays(cams, xo, xo)
        lengths, _ = O.depth_samples(S, 2.0, 0.0, None, hw)
        pts = O.ray_points(rays, lengths)
        grid = O.sample_grid(cams, pts)
        x0, y0, _, _, m = O.bilinear_corners(grid, r)
        assert torch.equal(res["points"].cpu(), pts) and torch.equal(res["grid"].cpu(), grid)
        assert torch.equal(res["x0"].cpu(), x0) and torch.equal(res["y0"].cpu(), y0) and torch.equal(res["mask"].cpu(), m)
        assert torch.equal(ops.patch_rays(cams.to(DEV), xs, xs).cpu(), rays)
    # the two targets differ (the path moved the camera), the references do not
    assert not torch.equal(pack_cameras(batches[0]["pose"])[0, 0], pack_cameras(batches[1]["pose"])[0, 0])
    assert torch.equal(pack_cameras(batches[0]["pose"])[0, 1:], pack_cameras(batches[1]["pose"])[0, 1:])


@torch.no_grad()
def test_f2_fused_sampler_step_on_the_gpu_walks_the_reference_trajectory():
    """f2: EulerEDMSampler + ScheduledCFGImgTextRef + DiscreteDenoiser(EpsScaling, LegacyDDPM) as the product launches them per denoise
    step (cd360.sampler.fused_cfg3_euler_step: sigma snapped on the device, c_in, network, then cd360_cfg_euler_step_f32 for c_out +
    3-way CFG + to_d + Euler) ON THE GPU over the 12-step trajectory of tests/golden/sampler.npz -- the REFERENCE's own classes around the
    same deterministic dummy network (tests/golden/make_golden.py::case_sampler).  Compared with the golden, not with the un-fused chain."""
    from test_sampler_cpu import load, run_product_steps
    g = load()
    got = run_product_steps(g, DEV, fused=True)
    assert got.is_cuda
    err = float((got.cpu() - g["cfg3"]).abs().max())
    print("f2: fused GPU trajectory vs the reference's golden, max abs:", err, "of max", float(g["cfg3"].abs().max()))
    assert torch.allclose(got.cpu(), g["cfg3"], atol=2e-5, rtol=1e-5), err
    # the reference's module stack on the GPU (un-fused, same classes the YAML instantiates) agrees too
    from cd360 import sampler as S
    den = S.DiscreteDenoiser().to(DEV)
    smp = S.EulerEDMSampler(num_steps=50, guider_config={"target": "sgm.modules.diffusionmodules.guiders.ScheduledCFGImgTextRef",
                                                         "params": {"scale": 7.5, "scale_im": 3.5}}, device=DEV)
    from test_sampler_cpu import dummy_network
    c = {"crossattn": g["c_crossattn"].to(DEV), "vector": g["c_vector"].to(DEV)}
    uc = {"crossattn": g["uc_crossattn"].to(DEV), "vector": g["uc_vector"].to(DEV)}
    res, _ = smp(lambda inp, s, cc: den(dummy_network, inp, s, cc), g["x"].to(DEV), c, uc=uc, num_steps=12)
    assert torch.allclose(res.cpu(), g["cfg3"], atol=2e-5, rtol=1e-5)


@torch.no_grad()
def test_f2_stage_kernels_of_the_captured_step_match_the_module_arithmetic():
    """The two ends of a captured sampling step (cd360_unet_stage_in / cd360_cfg_euler_step_cl, round 6) against the module chain they
    replace: DiscreteDenoiser.network_inputs' c_in scaling + the UNet's 4 -> 320 input convolution on the bf16 latent (fp32 torch
    restatement of the same roundings), silu(time_embed row + label_emb row), and cd360_cfg_euler_step_f32's update on the fp32 copy of
    bf16 channels-last eps rows -- with the scalars read through the device-side step index."""
    from cd360 import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    bs, rep, H, W, cout, E, nsteps = 2, 3, 24, 40, 320, 1280, 5   # (W is not a multiple of the kernel's 64-pixel row tile)
    x = torch.randn(bs, 4, H, W, generator=g, device=DEV)
    w = (torch.randn(cout, 4, 3, 3, generator=g, device=DEV) * 0.2).to(BF)
    bias = torch.randn(cout, generator=g, device=DEV)
    tab = torch.rand(nsteps, 4, generator=g, device=DEV) + 0.5
    temb = torch.randn(nsteps, E, generator=g, device=DEV).to(BF)
    lab = torch.randn(rep * bs, E, generator=g, device=DEV).to(BF)
    h = torch.empty(rep * bs, H * W, cout, dtype=BF, device=DEV)
    act = torch.empty_like(lab)
    for step in (0, 3):
        gi = torch.tensor([step], dtype=torch.int32, device=DEV)
        ops.unet_stage_in(x, tab, gi, w.float().permute(2, 3, 1, 0).reshape(36, cout).contiguous(), bias, temb, lab, h, act)
        x_in = (x * tab[step, 2]).to(BF).float()
        want = torch.nn.functional.conv2d(x_in.double(), w.double(), bias.double(), padding=1).float()  # [bs, cout, H, W]
        want = want.permute(0, 2, 3, 1).reshape(bs, H * W, cout)
        for r in range(rep):
            assert rel(h[r * bs:(r + 1) * bs], want) < 4e-3  # one bf16 rounding of the output
            assert torch.equal(h[r * bs:(r + 1) * bs], h[:bs])
        emb = (temb[step][None] + lab)  # bf16 + bf16 -> bf16, as `emb = time_embed(..) + label_emb(y)` in UNetModel.forward
        assert rel(act, torch.nn.functional.silu(emb)) < 4e-3
        # tail: in place on x against the fp32 kernel fed the same rows
        eps16 = torch.randn(3 * bs, H * W, 16, generator=g, device=DEV).to(BF)
        eps_nchw = eps16[..., :4].float().reshape(3 * bs, H, W, 4).permute(0, 3, 1, 2).contiguous()
        want_x = ops.cfg_euler_step(x, eps_nchw, tab[step, 0].reshape(1).contiguous(), tab[step, 1].reshape(1).contiguous(), 7.5, 3.5)
        x2 = x.clone()
        out = ops.cfg_euler_step_cl(x2, eps16[..., :4], tab, gi, 7.5, 3.5)
        assert out is x2 and torch.equal(x2, want_x)
