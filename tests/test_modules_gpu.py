"""The drop-in modules (sgm.modules.* on the HIP kernels, bf16) against the golden vectors the reference itself produced
(tests/golden/*.npz) and, at BASELINE.json sizes where the CPU oracle would take too long, through size-independent
properties of the path.  Needs an MI355X."""
import gzip
import json
import os

import numpy as np
import pytest
import torch

import weights as W
from cd360.cameras import unpack_cameras

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF = torch.bfloat16
# Bars of the toy-width module goldens (C = 64 / 128, written by the reference in fp32 on fp32 weights; the modules hold bf16 weights and
# round their activations to bf16 between kernels).  Each bar is the measurement of round 5 (printed per test when the module is done) plus
# a margin: one block 7.9e-3 -> the north star's 1e-2; five blocks / the reduced UNet 1.25e-2 ... 1.36e-2 -> 1.6e-2; the mask_ref goldens
# (features zeroed per view by a 0/1 mask: small maxima, same absolute errors) 2.24e-2 -> 2.5e-2.
TOL_BLOCK = 1e-2
TOL_DEEP = 1.6e-2
TOL = 2.5e-2


def load(name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, name + ".npz")).items()}


_WORST = {}  # test id -> largest rel() it evaluated: printed when the module is done, so that every bar below can be read against its measurement


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    r = (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    _WORST[test] = max(_WORST.get(test, 0.0), r)
    return r


@pytest.fixture(scope="module", autouse=True)
def _print_worst_rel():
    yield
    print("\nlargest max-norm relative error evaluated per test (tests/test_modules_gpu.py):")
    for k, v in sorted(_WORST.items()):
        print(f"  {k}: {v:.3e}")


def elem(got, want, rtol=1e-2, floor=1e-3):
    """Element-wise form of the parity bar: max over elements of |got - want| / (rtol |want| + floor max|want|); <= 1 means EVERY element is
    within `rtol` of its own reference value, with an absolute floor of `floor` of the tensor's maximum for the entries near zero.  `rel`
    above (max-norm) is how this repository reads north_star's "within 1e-2 relative" (DESIGN.md section 2); for tensors that hold many
    entries near 0 beside a few near 1 (alphas, fg) the max-norm alone would hide per-element errors of any size below 1 % of the
    maximum, so the render outputs are held to this form as well."""
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    return ((got - want).abs() / (rtol * want.abs() + floor * want.abs().max().clamp_min(1e-12))).max().item()


def dev(x):
    return x.to(DEV, BF)


def make_block(seed, C=64, heads=1, cd=32, S=4):
    from sgm.modules.attention import BasicTransformerBlock
    blk = BasicTransformerBlock(C, heads, 64, context_dim=cd, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=S, rgb_predict=True, mode="feature-nerf", stratified=True).eval()
    W.load_into(blk, seed=seed)
    return blk.to(DEV, BF)


def make_st(seed):
    from sgm.modules.attention import SpatialTransformer
    st = SpatialTransformer(128, 2, 64, depth=5, context_dim=32, use_linear=True, attn_type="softmax-xformers", use_checkpoint=False,
                            image_cross=True, rgb_predict=True, far=2, num_samples=4, mode="feature-nerf", stratified=True).eval()
    W.load_into(st, seed=seed)
    return st.to(DEV, BF)


@torch.no_grad()
def test_pose_block_matches_reference_golden():
    g = load("block_eval")
    blk = make_block(2)
    pose = unpack_cameras(g["cams"])
    out, fg, wts, alphas, rgb = blk(dev(g["x"]), context=dev(g["ctx"]), context_ref=dev(g["cref"]), pose=pose)
    assert wts is None
    assert rel(out, g["out"]) < TOL_BLOCK and rel(fg, g["fg"]) < TOL_BLOCK and rel(alphas, g["alphas"]) < TOL_BLOCK and rel(rgb, g["rgb"]) < TOL_BLOCK
    assert rel(blk(dev(g["x"]), context=dev(g["ctx"]))[0], g["plain"]) < TOL_BLOCK


@torch.no_grad()
@pytest.mark.parametrize("C,heads", [(640, 10), (1280, 20)])
def test_pose_block_at_sdxl_width_matches_reference_golden(C, heads):
    """The HIP pose block at the shipped config's widths (10 / 20 heads of 64, text context 2048 x 77) against the REFERENCE's own block
    on the same inputs (tests/golden/block_sdxl.npz from make_golden.py::case_block_sdxl; the oracle is pinned on the same fixture in
    tests/test_oracle_cpu.py).  The reference computes in fp32 on fp32 weights; the module holds bf16 weights and rounds its
    activations to bf16 between kernels: every output -- the render's fg / alphas / rgb and the whole block with and without the pose
    path -- inside the 1e-2 bar of the north star."""
    from make_golden_params import sdxl_block_inputs
    from sgm.modules.attention import BasicTransformerBlock
    g = load("block_sdxl")
    blk = BasicTransformerBlock(C, heads, 64, context_dim=2048, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=4, rgb_predict=True, mode="feature-nerf", stratified=True).eval()
    W.load_into(blk, seed=6)
    blk = blk.to(DEV, BF)
    x, ctx, cref, pose = sdxl_block_inputs(C)
    out, fg, wts, alphas, rgb = blk(dev(x), context=dev(ctx), context_ref=dev(cref), pose=pose)
    errs = {"out": rel(out, g[f"c{C}_out"]), "fg": rel(fg, g[f"c{C}_fg"]), "alphas": rel(alphas, g[f"c{C}_alphas"]), "rgb": rel(rgb, g[f"c{C}_rgb"]),
            "plain": rel(blk(dev(x), context=dev(ctx))[0], g[f"c{C}_plain"])}
    print(f"SDXL-width pose block C = {C} vs the reference's golden:", {k: round(v, 5) for k, v in errs.items()})
    assert wts is None
    assert max(errs.values()) < 1e-2, errs  # measured: out 7.5e-3 / 6.2e-3, plain 6.6e-3 / 6.5e-3, fg / alphas / rgb <= 8.4e-4


@torch.no_grad()
def test_fused_linear_route_equals_library_route(monkeypatch):
    """The inference path on cd360_gemm_bf16 (LayerNorm folded into the GEMM epilogues, GEGLU / residual / row statistics fused) against
    the same modules on the library GEMM + separate LayerNorm / GEGLU kernels (CD360_LIBRARY_LINEAR=1): same bf16 tensors in, so the
    two only differ by where intermediate values are rounded."""
    g = load("st_dual")
    st = make_st(3)
    pose = unpack_cameras(g["cams"])
    args = (dev(g["x"]), dev(g["xr"]))
    kw = dict(context=dev(g["ctx"]), contextr=dev(g["ctxr"]), pose=pose)
    assert st._fused_route(args[0])
    fused = st(*args, **kw)
    from cd360 import routes
    monkeypatch.setattr(routes, "library_linear", True)
    assert not st._fused_route(args[0])
    lib = st(*args, **kw)
    # five blocks deep, bf16 residual stream: the routes round intermediates at different places (1.2e-2 measured)
    assert rel(fused[0], lib[0]) < 2e-2 and rel(fused[1], lib[1]) < 2e-2
    for a, b in zip(fused[2] + fused[4] + fused[5], lib[2] + lib[4] + lib[5]):
        assert rel(a, b) < 1e-2
    # ... and the fused route is no further from the reference's fp32 golden than the library route
    assert rel(fused[0], g["out"]) < max(TOL, 1.25 * rel(lib[0], g["out"]))


@torch.no_grad()
def test_nerf_module_matches_reference_golden():
    from sgm.modules.nerfsd_pytorch3d import NerfSDModule
    g = load("nerf_eval")
    m = NerfSDModule(mode="feature-nerf", out_channels=64, far_plane=2.0, num_samples=4, rgb_predict=True, stratified=True).eval()
    W.load_into(m, seed=1)
    m = m.to(DEV, BF)
    feats, sigma, dists, vw, rgb, a, b = m(unpack_cameras(g["cams"]), dev(g["xref"]))
    assert a is None and b is None
    assert rel(feats, g["feats"]) < 1e-2 and rel(sigma, g["sigma"]) < 1e-2 and rel(rgb, g["rgb"]) < 1e-2 and rel(vw, g["view_weights"]) < 1e-2
    assert torch.equal(dists.cpu(), g["dists"])
    rays, pts, d2, _, _ = m.raymarcher(unpack_cameras(g["cams"]), 8, None, device=DEV)
    assert torch.equal(rays.cpu(), g["rays"]) and torch.equal(pts.cpu(), g["points"])


@torch.no_grad()
def test_nerf_module_average_views_matches_oracle():
    """FeatureNeRFEncoding(average=True) (nerfsd_pytorch3d.py:156-158: the per-view MLP outputs are averaged instead of weighted by the
    `nviews` softmax) -- an option the shipped config leaves off; served by the same fused kernel with zero view logits.  Against the
    oracle's `average=True` branch on the inputs of the eval golden; the view weights are None, as in the reference."""
    from oracle import pose_path as O
    from sgm.modules.nerfsd_pytorch3d import NerfSDModule
    g = load("nerf_eval")
    m = NerfSDModule(mode="feature-nerf", out_channels=64, far_plane=2.0, num_samples=4, rgb_predict=True, stratified=True, average=True).eval()
    w = {k: v.to(BF).float() for k, v in W.load_into(m, seed=1).items()}
    m = m.to(DEV, BF)
    feats, sigma, dists, vw, rgb, _, _ = m(unpack_cameras(g["cams"]), dev(g["xref"]))
    want = O.nerf_module(O.sub(w, "model"), g["cams"], g["xref"].to(BF).float(), 4, 2.0, average=True)
    assert vw is None and want[3] is None
    assert rel(feats, want[0]) < 1e-2 and rel(sigma, want[1]) < 1e-2 and rel(rgb, want[4]) < 1e-2
    # and it is a different function from the softmax-weighted one
    m2 = NerfSDModule(mode="feature-nerf", out_channels=64, far_plane=2.0, num_samples=4, rgb_predict=True, stratified=True).eval()
    W.load_into(m2, seed=1)
    assert rel(m2.to(DEV, BF)(unpack_cameras(g["cams"]), dev(g["xref"]))[0], want[0]) > 2e-2


@torch.no_grad()
def test_importance_sampled_depths_match_the_oracle():
    """SURVEY section 8 row f4 (dead upstream, F3): Raymarcher.importance_sampling on cd360_sample_pdf against the oracle's restatement --
    the recorded inputs of the reference's own call (tests/golden/importance_sampling.npz), both grid sizes, and the FeatureNeRF
    evaluated at those per-ray depths (NerfSDModule.forward(prev_weights=...)) against the oracle at the same depths."""
    from oracle import pose_path as O
    from sgm.modules.nerfsd_pytorch3d import NerfSDModule, Raymarcher
    gi = np.load(os.path.join(GOLD, "importance_sampling.npz"))
    far, near = float(gi["far"]), float(gi["near"])
    for tag, num_rays in (("same", 16), ("resized", 64)):
        pw = torch.from_numpy(gi[f"{tag}_prev_weights"])
        S = pw.shape[2]
        rm = Raymarcher(num_samples=S, far_plane=near + far, stratified=False, training=False, near_plane=near).to(DEV)
        t, d = rm.importance_sampling(pw.to(DEV), num_rays, S, DEV)
        wt, wd = O.importance_sampling(pw, num_rays, S, far, near)
        # the 0.01 floor keeps every bin's mass above ~1e-3: the last bit of the cdf moves a sample by < 1e-4 (see test_sample_pdf)
        assert (t.cpu() - wt).abs().max() < 1e-4 and (d.cpu() - wd).abs().max() < 2e-4
        assert (t[..., 1:] >= t[..., :-1]).all() and (t >= near).all() and (t <= near + near + far).all()
    g = load("nerf_eval")
    S, hw = 4, 64
    m = NerfSDModule(mode="feature-nerf", out_channels=64, far_plane=2.0, num_samples=S, rgb_predict=True, stratified=True).eval()
    m.raymarcher.training = False
    w = {k: v.to(BF).float() for k, v in W.load_into(m, seed=1).items()}
    m = m.to(DEV, BF)
    pw = torch.rand(2, hw, S, 1, generator=torch.Generator().manual_seed(8)) ** 2
    pose, xref = unpack_cameras(g["cams"]), g["xref"].to(BF).float()
    feats, sigma, dists, vw, rgb, su, du = m(pose, dev(g["xref"]), prev_weights=pw.to(DEV))
    want = O.nerf_module(O.sub(w, "model"), g["cams"], xref, S, 2.0, prev_weights=pw)
    assert su is None and du is None  # as upstream: imp_sample_next_step is dropped unless honour_imp_sample_next_step
    assert dists.shape == (2, hw, S, 1) and (dists.cpu() - want[2]).abs().max() < 2e-4
    assert rel(feats, want[0]) < 1e-2 and rel(sigma, want[1]) < 1e-2 and rel(rgb, want[4]) < 1e-2 and rel(vw, want[3]) < 1e-2
    # the depths are not the uniform ones, and the uniform pass returns what the next block would sample from
    assert rel(m(pose, dev(g["xref"]))[0], want[0]) > 2e-2
    m.honour_imp_sample_next_step = True
    out = m(pose, dev(g["xref"]), prev_weights=pw.to(DEV), imp_sample_next_step=True)
    want_u = O.nerf_module(O.sub(w, "model"), g["cams"], xref, S, 2.0, prev_weights=pw, uniform_pass=True)[5]
    assert rel(out[5], want_u["sigma_uniform"]) < 1e-2 and torch.equal(out[6].cpu().expand_as(want_u["dists_uniform"]), want_u["dists_uniform"])
    assert torch.equal(out[0], feats)


@torch.no_grad()
def test_pose_blocks_chain_importance_sampling_when_revived():
    """attention.py:571-598, 849-858 with use_prev_weights_imp_sample: block k renders at depths drawn from block k-1's weights at the
    uniform depths.  Upstream the chain never starts (F3); with NerfSDModule.honour_imp_sample_next_step the module route carries it.
    Block level against the oracle (reference_attn with prev_weights / uniform pass), then a SpatialTransformer whose pose blocks chain."""
    from oracle import pose_path as O
    from sgm.modules.attention import BasicTransformerBlock, SpatialTransformer
    g = load("block_eval")
    C, heads, cd, S = 64, 1, 32, 4
    blk = BasicTransformerBlock(C, heads, 64, context_dim=cd, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=S, rgb_predict=True, mode="feature-nerf", stratified=True, use_prev_weights_imp_sample=True,
                                imp_sample_next_step=True).eval()
    blk.pose_featurenerf.raymarcher.training = False
    w = {k: v.to(BF).float() for k, v in W.load_into(blk, seed=2).items()}
    blk = blk.to(DEV, BF)
    pose = unpack_cameras(g["cams"])
    x, ctx, cref = dev(g["x"]), dev(g["ctx"]), dev(g["cref"])
    b, n = g["cams"].shape[0], g["cams"].shape[1] - 1
    hw = cref.shape[-2]
    # default = the reference: nothing is handed on, the result is the golden's
    out = blk(x, context=ctx, context_ref=cref, pose=pose)
    assert out[2] is None and rel(out[0], g["out"]) < TOL
    blk.pose_featurenerf.honour_imp_sample_next_step = True
    cref4 = g["cref"].to(BF).float().reshape(b, n, hw, C)
    first = blk(x, context=ctx, context_ref=cref, pose=pose)
    want1 = O.reference_attn(w, cref4, g["ctx"].to(BF).float(), g["cams"], heads, S, 2.0, uniform_pass=True)
    assert first[2] is not None and first[2].shape == (b, hw, S, 1) and rel(first[2], want1[4]["weights_uniform"]) < 1e-2
    assert rel(first[1], want1[1]) < TOL and rel(first[3], want1[2]) < TOL
    second = blk(x, context=ctx, context_ref=cref, pose=pose, prev_weights=first[2])
    want2 = O.reference_attn(w, cref4, g["ctx"].to(BF).float(), g["cams"], heads, S, 2.0, prev_weights=want1[4]["weights_uniform"],
                             uniform_pass=True)
    assert rel(second[1], want2[1]) < TOL and rel(second[3], want2[2]) < TOL and rel(second[4], want2[3]) < TOL
    assert rel(second[2], want2[4]["weights_uniform"]) < 1e-2
    assert rel(second[3], first[3]) > 1e-3  # the sampled depths moved
    # SpatialTransformer: depth 5, pose blocks at 0 and 4 -> the second samples from the first's weights
    g2 = load("st_dual")
    st = SpatialTransformer(128, 2, 64, depth=5, context_dim=32, use_linear=True, attn_type="softmax-xformers", use_checkpoint=False,
                            image_cross=True, rgb_predict=True, far=2, num_samples=4, mode="feature-nerf", stratified=True,
                            use_prev_weights_imp_sample=True).eval()
    W.load_into(st, seed=3)
    st = st.to(DEV, BF)
    args = (dev(g2["x"]), dev(g2["xr"]))
    kw = dict(context=dev(g2["ctx"]), contextr=dev(g2["ctxr"]), pose=unpack_cameras(g2["cams"]))
    plain = st(*args, **kw)
    assert plain[3] is None
    flags = [blk_.imp_sample_next_step for blk_ in st.transformer_blocks]
    for blk_ in st.transformer_blocks:
        if blk_.image_cross:
            blk_.pose_featurenerf.honour_imp_sample_next_step = True
            blk_.pose_featurenerf.raymarcher.training = False
    chained = st(*args, **kw)
    assert any(flags) and all(torch.isfinite(t).all() for t in (chained[0], chained[1]))
    assert rel(chained[2][0], plain[2][0]) < 1e-6  # the first pose block has nothing to sample from
    if flags[-1] is False and len(chained[2]) > 1:
        assert rel(chained[2][-1], plain[2][-1]) > 1e-4  # the last one rendered at importance-sampled depths


@torch.no_grad()
def test_mask_ref_forward_matches_reference_golden():
    """mask_ref (nerfsd_pytorch3d.py:61-70): NerfSDModule, the pose block and the tiny UNet with reference-view masks against the vectors
    the imported reference produced (tests/golden/mask_ref.npz)."""
    from make_golden_params import UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    from sgm.modules.nerfsd_pytorch3d import NerfSDModule
    g = load("mask_ref")
    m = NerfSDModule(mode="feature-nerf", out_channels=64, far_plane=2.0, num_samples=4, rgb_predict=True, stratified=True).eval()
    W.load_into(m, seed=1)
    m = m.to(DEV, BF)
    m.return_view_weights = True
    feats, sigma, dists, attn, rgb, _, _ = m(unpack_cameras(g["nerf_cams"]), dev(g["nerf_xref"]), mask_ref=g["nerf_mask"].to(DEV))
    assert rel(feats, g["nerf_feats"]) < TOL and rel(sigma, g["nerf_sigma"]) < TOL and rel(rgb, g["nerf_rgb"]) < TOL
    assert rel(attn, g["nerf_view_weights"]) < TOL
    blk = make_block(2)
    out, fg, wts, alphas, rgb = blk(dev(g["blk_x"]), context=dev(g["blk_ctx"]), context_ref=dev(g["blk_cref"]), pose=unpack_cameras(g["blk_cams"]),
                                    mask_ref=g["blk_mask"].to(DEV))
    assert rel(out, g["blk_eval_out"]) < TOL and rel(fg, g["blk_eval_fg"]) < TOL and rel(alphas, g["blk_eval_alphas"]) < TOL
    assert rel(rgb, g["blk_eval_rgb"]) < TOL
    plain_golden = load("block_eval")  # same weights / inputs without the mask: the masked run must be the closer one by far
    assert rel(fg, plain_golden["fg"]) > 4 * rel(fg, g["blk_eval_fg"]) and rel(rgb, plain_golden["rgb"]) > 4 * rel(rgb, g["blk_eval_rgb"])
    u = load("unet_tiny")
    net = UNetModel(**UNET_TINY).eval()
    W.load_into(net, seed=5)
    net = net.to(DEV, BF)
    eps, fgs, _, rgbs = net(u["x"].to(DEV), timesteps=u["t"].to(DEV), context=u["ctx"].to(DEV), y=u["y"].to(DEV), pose=unpack_cameras(u["cams"]),
                            input_ref=u["input_ref"].to(DEV), sigmas_ref=u["sigmas_ref"].to(DEV), mask_ref=g["unet_mask"].to(DEV))
    assert rel(eps, g["unet_out"]) < 4e-2
    for i in range(3):
        assert rel(fgs[i], g[f"unet_fg{i}"]) < 4e-2 and rel(rgbs[i], g[f"unet_rgb{i}"]) < 4e-2


@torch.no_grad()
def test_spatial_transformer_dual_stream_matches_reference_golden():
    g = load("st_dual")
    st = make_st(3)
    pose = unpack_cameras(g["cams"])
    out, xr, fgs, pw, alphas, rgbs = st(dev(g["x"]), dev(g["xr"]), context=dev(g["ctx"]), contextr=dev(g["ctxr"]), pose=pose)
    assert pw is None and len(fgs) == 2
    assert rel(out, g["out"]) < TOL_DEEP and rel(xr, g["xr_out"]) < TOL_DEEP
    for i in range(2):
        assert rel(fgs[i], g[f"fg{i}"]) < TOL_DEEP and rel(alphas[i], g[f"alphas{i}"]) < TOL_DEEP and rel(rgbs[i], g[f"rgb{i}"]) < TOL_DEEP
    assert rel(st(dev(g["x"]), None, context=dev(g["ctx"]))[0], g["plain"]) < TOL_DEEP


@torch.no_grad()
def test_native_reference_sampling_matches_sample_py_golden():
    """cd360.sampling.enable_reference_sampling == sample.py's monkey patch: render on step 0, cached render afterwards."""
    from cd360 import sampling
    g = load("customforward_cfg3")
    st = make_st(4)
    refs = {f"transformer_blocks.{d}": dev(W.tensor(f"references.{d}", (5, 64, 128), seed=4)) for d in (0, 4)}
    sampling.set_references(st, refs)
    assert sampling.enable_reference_sampling(st, g["choices"].tolist()) == ["transformer_blocks.0", "transformer_blocks.4"]
    pose = unpack_cameras(g["cams"])
    out0, xr, fgs, _, alphas, rgbs = st(dev(g["x0"]), None, context=dev(g["ctx"]), pose=pose)
    assert xr is None
    assert rel(out0, g["out0"]) < TOL_DEEP
    assert rel(st.transformer_blocks[0].rendered_feat, g["rend0"]) < TOL_DEEP and rel(st.transformer_blocks[4].rendered_feat, g["rend4"]) < TOL_DEEP
    assert rel(fgs[0], g["fg0"]) < TOL_DEEP and rel(rgbs[1], g["rgb1"]) < TOL_DEEP
    out1 = st(dev(g["x1"]), None, context=dev(g["ctx"]), pose=pose)[0]
    assert rel(out1, g["out1"]) < TOL_DEEP
    sampling.clear_rendered_feat(st)
    assert st.transformer_blocks[0].rendered_feat is None


@torch.no_grad()
def test_sample_py_style_rebinding_runs_the_fused_route_and_matches_sample_py_golden():
    """The UNCHANGED driver: `forward` rebound on the SpatialTransformer and on every block the way sample.py:247-262 does it (stand-in
    functions named and shaped like sample.py's, which raise if executed: tests/golden/sample_py_stub.py), `references` registered as the
    delta checkpoint loader does (sgm/util.py:231-235), the global `choices` set afterwards -- and NO call into cd360.sampling.  The
    rebinding is recognised, the fused route serves it (render on the first call, cached render on the second), and the outputs are the
    ones sample.py's OWN functions produced on the reference's modules (tests/golden/customforward_cfg3.npz); clearing `rendered_feat`
    by plain attribute assignment, as DiffusionEngine.clear_rendered_feat does (diffusion.py:165-170), renders again."""
    import sample_py_stub as SP
    g = load("customforward_cfg3")
    st = make_st(4)
    for d in (0, 4):
        st.transformer_blocks[d].register_buffer("references", dev(W.tensor(f"references.{d}", (5, 64, 128), seed=4)))
    SP.register(st, g["choices"].tolist())
    assert st._fused_route(dev(g["x0"])), "the recognised rebinding must leave the fused route open"
    pose = unpack_cameras(g["cams"])
    out0, xr, fgs, _, alphas, rgbs = st(dev(g["x0"]), None, context=dev(g["ctx"]), pose=pose)
    assert xr is None
    assert rel(out0, g["out0"]) < TOL_DEEP
    assert rel(st.transformer_blocks[0].rendered_feat, g["rend0"]) < TOL_DEEP and rel(st.transformer_blocks[4].rendered_feat, g["rend4"]) < TOL_DEEP
    assert rel(fgs[0], g["fg0"]) < TOL_DEEP and rel(rgbs[1], g["rgb1"]) < TOL_DEEP
    kept = st.transformer_blocks[0].rendered_feat
    out1 = st(dev(g["x1"]), None, context=dev(g["ctx"]), pose=pose)[0]
    assert rel(out1, g["out1"]) < TOL_DEEP and st.transformer_blocks[0].rendered_feat is kept  # cached render reused (sample.py:122-124)
    # ... bit-identical to the native switch (cd360.sampling.enable_reference_sampling) on a second instance
    from cd360 import sampling
    st2 = make_st(4)
    sampling.set_references(st2, {f"transformer_blocks.{d}": dev(W.tensor(f"references.{d}", (5, 64, 128), seed=4)) for d in (0, 4)})
    sampling.enable_reference_sampling(st2, g["choices"].tolist())
    assert torch.equal(st2(dev(g["x0"]), None, context=dev(g["ctx"]), pose=pose)[0], out0)
    for m in st.modules():  # DiffusionEngine.clear_rendered_feat
        if hasattr(m, "pose_emb_layers"):
            m.rendered_feat = None
    out0b = st(dev(g["x0"]), None, context=dev(g["ctx"]), pose=pose)[0]
    assert torch.equal(out0b, out0) and st.transformer_blocks[0].rendered_feat is not kept
    # a single block entered through __call__ (a driver that patches blocks only) is served the same way
    blk = st.transformer_blocks[0]
    blk.rendered_feat = None
    xt = dev(W.tensor("tok", (3, 64, 128), seed=4))
    o = blk(xt, context=dev(g["ctx"]), context_ref=xt, pose=pose)
    assert blk.rendered_feat is not None and rel(blk.rendered_feat, g["rend0"]) < TOL_DEEP and torch.isfinite(o[0]).all()


@torch.no_grad()
def test_unet_dual_stream_matches_reference_golden():
    from make_golden_params import UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    g = load("unet_tiny")
    net = UNetModel(**UNET_TINY).eval()
    W.load_into(net, seed=5)
    net = net.to(DEV, BF)
    out, fgs, alphas, rgbs = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), y=g["y"].to(DEV),
                                 pose=unpack_cameras(g["cams"]), input_ref=g["input_ref"].to(DEV), sigmas_ref=g["sigmas_ref"].to(DEV), mask_ref=None)
    assert out.dtype == torch.float32 and len(fgs) == 3 and len(alphas) == 3 and len(rgbs) == 3
    assert rel(out, g["out"]) < TOL_DEEP  # measured 1.34e-2 (round 5); 4e-2 until then
    for i in range(3):
        assert rel(fgs[i], g[f"fg{i}"]) < TOL_DEEP and rel(rgbs[i], g[f"rgb{i}"]) < TOL_DEEP


@torch.no_grad()
def test_sdxl_width_pose_block_properties():
    """Level-2 SDXL pose block (C=1280, 20 heads, r=32, S=24) with n=8 references: too big for the CPU oracle, so check
    properties: volume-render weights are a sub-probability (0 <= fg <= 1), alphas in [0,1], the render is independent of x,
    the block output depends on the reference features, and a repeated call is bit-identical (no races)."""
    from cd360 import synth
    blk = make_block(6, C=1280, heads=20, cd=2048, S=24)
    b, n, hw = 1, 8, 1024
    pose = synth.pose_batch(b, n, seed=3)
    x = dev(W.tensor("x", (b, hw, 1280), seed=6))
    ctx = dev(W.tensor("ctx", (b, 77, 2048), seed=6))
    cref = dev(W.tensor("cref", (b * n, hw, 1280), seed=6))
    out, fg, _, alphas, rgb = blk(x, context=ctx, context_ref=cref, pose=pose)
    out2, fg2, _, alphas2, rgb2 = blk(x, context=ctx, context_ref=cref, pose=pose)
    assert torch.equal(out, out2) and torch.equal(fg, fg2) and torch.equal(rgb, rgb2)
    assert torch.isfinite(out.float()).all()
    assert float(fg.min()) >= -1e-5 and float(fg.max()) <= 1 + 1e-4
    assert float(alphas.min()) >= 0 and float(alphas.max()) <= 1 and float(rgb.min()) >= 0 and float(rgb.max()) <= 1 + 1e-4
    _, fg3, _, _, _ = blk(x * 0.5 + 1.0, context=ctx, context_ref=cref, pose=pose)
    assert torch.equal(fg, fg3)  # the FeatureNeRF render does not read x (attention.py:571-598)
    out4 = blk(x, context=ctx, context_ref=cref * 0.5, pose=pose)[0]
    assert not torch.allclose(out.float(), out4.float(), atol=1e-2)


@torch.no_grad()
def test_cfgB_level1_sampling_block_cached_equals_uncached():
    """BASELINE configs[1] shape for one level-1 pose block: C=640, r=64 (hw=4096), S=24, n=50 references, CFG batch 3.
    Too large for the CPU oracle, so size-independent properties of the sampling path (sample.py:82-136):
      * a step that reuses the cached render equals a step that re-renders (the render does not depend on x or the step);
      * the unconditional third (null image for every view) differs from the conditional thirds, which are identical;
      * render outputs are finite, fg in [0, 1]; re-running is bit-identical."""
    from cd360 import sampling, synth
    blk = make_block(11, C=640, heads=10, cd=2048, S=24)
    n_train, n, hw = 50, 50, 4096
    refs = dev(W.tensor("references", (n_train + 1, hw, 640), seed=11))
    sampling.set_references(blk, {"": refs})
    sampling.enable_reference_sampling(blk, list(range(n)))
    pose = synth.pose_batch(1, n, seed=9, n_train=n_train) * 3
    ctx1 = dev(W.tensor("ctx", (1, 77, 2048), seed=11))
    ctx = ctx1.expand(3, -1, -1).contiguous()
    x1 = dev(W.tensor("x", (1, hw, 640), seed=11))
    x = x1.expand(3, -1, -1).contiguous()
    out_a, fg, _, alphas, rgb = blk(x, context=ctx, context_ref=x, pose=pose)  # renders
    rend = blk.rendered_feat.clone()
    out_b = blk(x, context=ctx, context_ref=x, pose=pose)[0]  # cached
    assert torch.equal(out_a, out_b)
    sampling.clear_rendered_feat(blk)
    out_c = blk(x * 0.5, context=ctx, context_ref=x, pose=pose)[0]  # different x, re-render
    assert torch.equal(blk.rendered_feat, rend)  # render independent of x, bit-reproducible
    assert torch.isfinite(out_c.float()).all()
    assert float(fg.min()) >= -1e-5 and float(fg.max()) <= 1 + 1e-4 and float(alphas.min()) >= 0 and float(alphas.max()) <= 1
    # the two conditional thirds see identical inputs; hipBLASLt's stream-K GEMMs may round their rows differently, so "equal"
    # is asserted to bf16 round-off rather than bitwise
    assert rel(rend[1], rend[2]) < 2e-2 and rel(rend[0], rend[1]) > 5e-2
    assert rel(out_a[1], out_a[2]) < 2e-2


def _oracle_render_on_rays(w, cams, cref, ctx, heads, S, far, idx):
    """oracle.reference_attn (attention.py:571-598) restricted to the target rays `idx`: every step of the FeatureNeRF render, of the
    pose-token cross-attention and of the volume render is independent per ray, so a subset costs seconds at n = 50 views where the
    whole 64 x 64 ray grid would need ~100 GB of fp32 intermediates.  cams [1, n+1, 16], cref [1, n, hw, C] (ALL of every reference
    map: the bilinear gather reads the full maps), ctx [1, 77, cd].  Returns (xref [1, k, C], fg, alphas, rgb, debug) for the k rays."""
    from oracle import pose_path as O
    hw = cref.shape[2]
    r = int(round(hw ** 0.5))
    xs = O.patch_positions(r)
    rays = O.patch_rays(cams, xs, xs)[:, :, idx]
    lengths, dists = O.depth_samples(S, far, 0.0, None, len(idx))
    pts = O.ray_points(rays, lengths)
    out, _, dbg = O.feature_nerf(O.sub(w, "pose_featurenerf.model"), cams, cref, rays, pts)
    sigma, feats = out[..., -1:], out[..., :-1]
    rgb, feats = feats[..., -3:], feats[..., :-3]
    k, C = feats.shape[1], feats.shape[-1]
    tok = feats.reshape(1, k * S, C)
    tok = O.cross_attention(O.sub(w, "attn2"), O.layer_norm(w, "norm2", tok), ctx, heads) + tok
    rendered, fg, alphas, _, rgb_out = O.vol_render(tok.reshape(1, k, S, C), O.trunc_exp(sigma), dists.unsqueeze(-1), torch.sigmoid(rgb))
    dbg.update(points=pts)
    return rendered, fg, alphas, rgb_out, dbg


@torch.no_grad()
@pytest.mark.parametrize("level", [1, 2])
def test_cfgB_pose_block_render_matches_the_oracle_on_a_ray_subset(level):
    _cfgB_ray_subset(level, fp8=False)


@torch.no_grad()
@pytest.mark.parametrize("level", [1, 2])
def test_cfgB_pose_block_render_fp8_attention_tolerance_report(level):
    """BASELINE configs[4] at workload level against the oracle: the same block, rays and inputs as the test above with the pose-token and
    text cross-attention contractions on fp8 MFMA (cd360.routes.fp8_attn -> cd360_qproj_attn_fp8_bf16).  The rendered features are a
    residual around the attention output (tok + attn2(norm2(tok))), so the e4m3 error of the attention (6e-2 ... 1e-1 of ITS max on
    unit-scale inputs, tests/test_gemm_gpu.py) reaches them attenuated; printed per CFG branch and bounded at 5e-2 -- outside the bf16 bar,
    which is why bf16 stays the default arithmetic."""
    _cfgB_ray_subset(level, fp8=True)


def _cfgB_ray_subset(level, fp8):
    """BASELINE configs[1] -- the headline configuration -- pinned on the oracle: one pose block at the 640 level (r = 64: 4096 rays) and
    one at the 1280 level (r = 32), n = 50 reference views, S = 24 depth samples, CFG batch 3 = [null image | image | image + text] through
    the product's sampling route (references buffer, de-duplicated render, fused pose-token attention with Nq = hw * 24 per branch, render
    commuted with the out projection).  128 target rays spread over the image (all four borders included, where projections leave the
    reference maps) are rendered by the fp32 oracle at full n / r / S for each of the three branches -- on the SAME inputs: weights,
    references and text context rounded to bf16 on both sides -- and compared with the block's rendered features, foreground mask,
    alphas and rgb at those rays: the north star's 1e-2 bar.  The integer corner indices / in-bounds masks / grid coordinates of the same
    rays at all 50 views are bit-exact (cd360_ray_project_index against oracle.bilinear_corners)."""
    from cd360 import nerf, ops, routes, sampling, synth
    from cd360.cameras import pack_cameras
    from oracle import pose_path as O
    from sgm.modules.attention import BasicTransformerBlock
    C, heads, r = (640, 10, 64) if level == 1 else (1280, 20, 32)
    hw, S, n, n_train, cd = r * r, 24, 50, 50, 2048
    blk = BasicTransformerBlock(C, heads, 64, context_dim=cd, checkpoint=False, attn_mode="softmax-xformers", image_cross=True, far=2,
                                num_samples=S, rgb_predict=True, mode="feature-nerf", stratified=True).eval()
    w = {k: v.to(BF).float() for k, v in W.load_into(blk, seed=40 + level).items()}  # what the bf16 module holds, as fp32
    blk = blk.to(DEV, BF)
    refs = W.tensor("references", (n_train + 1, hw, C), seed=40 + level).to(BF)
    choices = [int(c) for c in np.random.default_rng(level).permutation(n_train)[:n]]
    sampling.set_references(blk, {"": refs.to(DEV)})
    sampling.enable_reference_sampling(blk, choices)
    pose = synth.pose_batch(1, n, seed=9, n_train=n_train) * 3
    ctx = W.tensor("ctx", (3, 77, cd), seed=40 + level).to(BF)
    x = dev(W.tensor("x", (3, hw, C), seed=40 + level))
    with routes.override(fp8_attn=fp8):
        out, fg, _, alphas, rgb = blk(x, context=ctx.to(DEV), context_ref=x, pose=pose)
    rend = blk.rendered_feat
    assert rend.shape == (3, hw, C) and fg.shape[:2] == (3, hw) and alphas.shape[:3] == (3, hw, S)
    # 128 rays: the four corners, points on every border, the rest spread over the interior
    g = np.random.default_rng(100 + level)
    border = [0, r - 1, hw - r, hw - 1] + [int(v) for v in g.integers(1, r - 1, 8)] + [int(v) * r for v in g.integers(1, r - 1, 8)] \
        + [int(v) * r + r - 1 for v in g.integers(1, r - 1, 8)] + [hw - r + int(v) for v in g.integers(1, r - 1, 8)]
    idx = torch.tensor(sorted(set(border) | set(int(v) for v in g.choice(hw, 128, replace=False)))[:128])
    cams = pack_cameras(pose[:1]).float().cpu()  # [1, n + 1, 16]
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    cond = refs[:-1][torch.tensor(choices)].float()[None]          # [1, n, hw, C]
    null = refs[-1:].float()[None].expand(1, n, -1, -1)           # the unconditional third: the null image for every view
    errs, elems, dbg = {}, {}, None
    for br in range(3):
        want = _oracle_render_on_rays(w, cams, null if br == 0 else cond, ctx[br:br + 1].float(), heads, S, 2.0, idx)
        dbg = want[4]
        pairs = ((rend[br:br + 1, idx.to(DEV)], want[0]), (fg[br:br + 1, idx.to(DEV)].reshape(want[1].shape), want[1]),
                 (alphas[br:br + 1, idx.to(DEV)].reshape(want[2].shape), want[2]), (rgb[br:br + 1, idx.to(DEV)].reshape(want[3].shape), want[3]))
        errs[br] = tuple(rel(a_, b_) for a_, b_ in pairs)
        elems[br] = tuple(elem(a_, b_) for a_, b_ in pairs)
    print(f"cfg-B level-{level} pose block ({'fp8' if fp8 else 'bf16'} attention) vs oracle on {len(idx)} rays (xref, fg, alphas, rgb) per CFG branch:",
          {k: tuple(round(e, 4) for e in v) for k, v in errs.items()})
    assert max(max(v) for v in errs.values()) < (5e-2 if fp8 else 1e-2), errs
    if fp8:
        return
    # element-wise: every fg / alphas / rgb entry within 1e-2 of ITS OWN oracle value (+ 1e-3 of the tensor maximum for entries near 0);
    # the rendered features are zero-centred sums over 24 samples (no scale of their own per element), so for them the floor is the bar
    print("   element-wise |d| / (1e-2 |want| + 1e-3 max|want|) (xref, fg, alphas, rgb):", {k: tuple(round(e, 3) for e in v) for k, v in elems.items()})
    assert max(max(v[1:]) for v in elems.values()) <= 1.0, elems
    # integer ray indices at the full camera set: bit-exact
    xs = nerf.patch_positions(r, DEV)
    t, _ = nerf.depth_samples(S, 2.0, 0.0, DEV, hw)
    res = ops.ray_project_index(cams.to(DEV), xs, xs, t)
    x0_o, y0_o, _, _, m_o = O.bilinear_corners(dbg["grid"], r)
    assert torch.equal(res["points"][:, idx.to(DEV)].cpu(), dbg["points"])
    assert torch.equal(res["grid"][:, :, idx.to(DEV)].cpu(), dbg["grid"])
    assert torch.equal(res["x0"][:, :, idx.to(DEV)].cpu(), x0_o) and torch.equal(res["y0"][:, :, idx.to(DEV)].cpu(), y0_o)
    assert torch.equal(res["mask"][:, :, idx.to(DEV)].cpu(), m_o) and int((m_o != 15).sum()) > 0


@torch.no_grad()
def test_cfg_branch_deduplication_is_bit_identical(monkeypatch):
    """3-way CFG hands the block `[pose] * 3`: the image-conditional and image+text-conditional thirds share pose and references,
    so the FeatureNeRF part is rendered for two thirds and reused for the third.  Must equal rendering all three, bit for bit
    (same kernels, same inputs), and must not trigger when the thirds are different camera objects."""
    from cd360 import sampling, synth
    from cd360.cameras import join_cameras_as_batch
    from cd360 import routes
    monkeypatch.setattr(routes, "no_cfg_dedup", False)  # (the A/B knob would switch the feature under test off)
    blk = make_block(13, C=128, heads=2, cd=32, S=6)
    n_train, n, hw = 6, 6, 256
    sampling.set_references(blk, {"": dev(W.tensor("references", (n_train + 1, hw, 128), seed=13))})
    sampling.enable_reference_sampling(blk, list(range(n)))
    one = synth.pose_batch(1, n, seed=3, n_train=n_train)
    same = one * 3
    distinct = [join_cameras_as_batch([one[0][i] for i in range(n + 1)]) for _ in range(3)]
    x = dev(W.tensor("x", (3, hw, 128), seed=13))
    ctx = dev(W.tensor("ctx", (3, 77, 32), seed=13))
    assert blk._duplicate_cfg_branch(same, (3, n, hw, 128)) == 1 and blk._duplicate_cfg_branch(distinct, (3, n, hw, 128)) == 0
    outs = []
    for pose in (same, distinct):
        sampling.clear_rendered_feat(blk)
        o = blk(x, context=ctx, context_ref=x, pose=pose)
        outs.append((o[0], o[1], o[3], o[4], blk.rendered_feat.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    assert not torch.equal(outs[0][4][1], outs[0][4][2])  # the two conditional thirds still differ after the text cross-attention


@torch.no_grad()
def test_cfg_deduplication_with_three_samples_per_step(monkeypatch):
    """Three diffusion samples per step (x batch 9 = [3 null | 3 image | 3 image+text]): the de-duplicated render batch has 6 elements,
    which must be read as 3 null + 3 conditional, not as a 3-way batch of 2 (the layout is passed explicitly, never inferred from the
    de-duplicated size).  Bit-identical to the path with the de-duplication switched off."""
    from cd360 import routes, sampling, synth
    blk = make_block(14, C=128, heads=2, cd=32, S=6)
    n_train, n, hw, bs = 6, 6, 256, 3
    sampling.set_references(blk, {"": dev(W.tensor("references", (n_train + 1, hw, 128), seed=14))})
    sampling.enable_reference_sampling(blk, list(range(n)))
    poses = synth.pose_batch(bs, n, seed=5, n_train=n_train)
    pose9 = poses * 3
    x = dev(W.tensor("x", (3 * bs, hw, 128), seed=14))
    ctx = dev(W.tensor("ctx", (3 * bs, 77, 32), seed=14))
    assert blk._duplicate_cfg_branch(pose9, (3 * bs, n, hw, 128)) == bs
    outs = []
    for off in (False, True):
        monkeypatch.setattr(routes, "no_cfg_dedup", off)
        sampling.clear_rendered_feat(blk)
        o = blk(x, context=ctx, context_ref=x, pose=pose9)
        outs.append((o[0], o[1], o[3], o[4], blk.rendered_feat.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    assert not torch.equal(outs[0][4][0], outs[0][4][bs])  # unconditional (null image) and conditional renders differ


@torch.no_grad()
def test_references_harvest_delta_checkpoint_and_sampling_round_trip():
    """§8 f3 on the HIP path: run the reference images through the UNet WITHOUT a pose with the harvest hooks on
    (diffusion.py:151-163), build `references` (main.py:596-607), save/load the delta checkpoint into a second UNet
    (main.py:611-624, sgm/util.py:227-240) and sample from it with the native reference-sampling mode (sample.py:82-136):
    both UNets must give the same step output, and the harvested rows must equal the blocks' plain outputs."""
    from cd360 import finetune, sampling, synth
    from make_golden_params import UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    src, dst = UNetModel(**UNET_TINY).eval(), UNetModel(**UNET_TINY).eval()
    W.load_into(src, seed=5)
    W.load_into(dst, seed=5)
    for n, p in dst.named_parameters():
        if "pose" in n:
            p.zero_()  # so that only the delta checkpoint can make dst agree with src
    src, dst = src.to(DEV, BF), dst.to(DEV, BF)
    n_train, L = 3, 16
    imgs = W.tensor("ft.refs", (n_train + 1, 4, L, L), seed=12).to(DEV)
    imgs[-1] = 0  # last row: the null image (data_co3d.py:469-472)
    ctx = W.tensor("ft.ctx", (1, 77, 32), seed=12).to(DEV)
    y = W.tensor("ft.y", (1, 16), seed=12).to(DEV)
    t = torch.full((1,), 10.0, device=DEV)
    acts, handles = finetune.register_reference_hooks(src)
    for i in range(n_train + 1):
        src(imgs[i:i + 1], timesteps=t, context=ctx, y=y)
    refs = finetune.harvest_references(src, acts)
    finetune.remove_hooks(handles)
    assert len(refs) == 3 and all(r.shape[0] == n_train + 1 and r.dtype == BF for r in refs.values())
    full = {"model.diffusion_model." + k: v for k, v in src.state_dict().items()}
    delta = finetune.delta_state_dict(full)
    assert finetune.load_delta_state_dict(dst, delta) == []
    pose = synth.pose_batch(1, n_train, seed=4, n_train=n_train) * 3
    x = W.tensor("ft.x", (3, 4, L, L), seed=12).to(DEV)
    outs = []
    for net in (src, dst):
        sampling.enable_reference_sampling(net, list(range(n_train)))
        o = net(x, timesteps=t.expand(3), context=ctx.expand(3, -1, -1).contiguous(), y=y.expand(3, -1).contiguous(), pose=pose)
        outs.append(o)
        sampling.clear_rendered_feat(net)
    assert torch.isfinite(outs[0][0]).all() and len(outs[0][1]) == 3
    assert rel(outs[1][0], outs[0][0]) < 1e-3 and rel(outs[1][3][0], outs[0][3][0]) < 1e-3
    assert rel(outs[0][0][0], outs[0][0][2]) > 1e-4  # conditional vs null-image third differ: the references are in use


@torch.no_grad()
def test_training_loss_on_hip_outputs_matches_loss_on_reference_outputs():
    """§8 f3: StandardDiffusionLossImgRef.get_loss (loss.py:177-209) evaluated on the HIP UNet's dual-stream outputs vs on the
    reference's own outputs for the same inputs (unet_tiny golden): every term within the module tolerance."""
    from make_golden_params import LOSS_CFG, UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    from sgm.util import instantiate_from_config
    g = load("unet_tiny")
    net = UNetModel(**UNET_TINY).eval()
    W.load_into(net, seed=5)
    net = net.to(DEV, BF)
    out, fgs, alphas, rgbs = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), y=g["y"].to(DEV),
                                 pose=unpack_cameras(g["cams"]), input_ref=g["input_ref"].to(DEV), sigmas_ref=g["sigmas_ref"].to(DEV), mask_ref=None)
    b, L = g["x"].shape[0], g["x"].shape[-1]
    target = W.tensor("lt.x0", (b, 4, L, L), seed=2).to(DEV)
    rgb_t = W.tensor("lt.rgb", (b, 3, 8 * L, 8 * L), seed=2).clamp(-1, 1).to(DEV)
    mask = torch.ones(b, 1, L, L, device=DEV)
    opacity = torch.sigmoid(3 * W.tensor("lt.op", (b, 1, 8 * L, 8 * L), seed=2)).to(DEV)
    w = torch.full((b, 1, 1, 1), 0.7, device=DEV)
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    got = loss_fn.get_loss(out, fgs, rgbs, target, rgb_t, w, mask, None, opacity, alphas)
    ref_f = [g[f"fg{i}"].to(DEV) for i in range(3)]
    ref_r = [g[f"rgb{i}"].to(DEV) for i in range(3)]
    want = loss_fn.get_loss(g["out"].to(DEV), ref_f, ref_r, target, rgb_t, w, mask, None, opacity, [a.float() for a in alphas])
    for a, bb in zip(got, want):
        assert a.dtype == torch.float32 and torch.isfinite(a).all() and rel(a, bb) < 4e-2
    assert got[1].shape == (b, 3) and got[3].shape == (b, 3)


@torch.no_grad()
def test_full_sdxl_unet_configA_matches_cpu_oracle():
    """BASELINE configs[0] (the reference's own CPU-runnable case) at FULL SDXL width and depth: 512^2 image = latent 64^2, b = 1,
    n = 4 reference views, dual-stream eval forward (openaimodel.py:975-1093) with all 12 FeatureNeRF renders, 70 transformer
    blocks, 2.6 B random parameters.  HIP bf16 path vs the fp32 CPU oracle (oracle/pose_path.py::unet_forward, itself pinned on the
    reference's golden vectors at reduced width): eps and every render output.  Takes about a minute of host time."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pose_path as O
    from cd360 import synth
    from cd360.cameras import pack_cameras
    from make_golden_params import SDXL_NETWORK_CONFIG
    from sgm.util import instantiate_from_config
    net = instantiate_from_config(SDXL_NETWORK_CONFIG).eval()
    sd = W.load_into(net, seed=21)
    net = net.to(DEV, BF)
    b, n, L = 1, 4, 64
    cams = pack_cameras(synth.pose_batch(b, n, seed=5))
    x = W.tensor("A.x", (b, 4, L, L), seed=21)
    xr = W.tensor("A.xr", (b, n, 4, L, L), seed=21)
    ctx = W.tensor("A.ctx", (b + b * n, 77, 2048), seed=21)
    y = W.tensor("A.y", (b + b * n, 2816), seed=21)
    t, tr = torch.tensor([500.0]), torch.tensor([3.0])
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    # Teacher forcing: record what EVERY unit of the oracle's target stream was given and what it returned -- all 70 transformer blocks
    # (12 of them pose blocks, with their render outputs), all ResBlocks, proj_in / proj_out of all 11 SpatialTransformers, the two
    # Downsample and the two Upsample convolutions -- so that each HIP unit can be judged on identical inputs, separately from the drift
    # the bf16 residual stream accumulates over 70 blocks.  (Identical inputs include the parameters: the oracle runs on the bf16-rounded
    # weights the HIP modules hold, in fp32 arithmetic.)
    sd = {k: v.to(BF).float() for k, v in sd.items()}
    orig = dict(block=O.transformer_block, res=O.res_block, st=O.spatial_transformer, seq=O._run_block)
    blocks_rec, res_rec, st_rec, seq_rec = [], [], [], {}

    def rec_block(w, xin, context, heads, context_ref=None, cams=None, rendered_feat=None, **kw):
        res = orig["block"](w, xin, context, heads, context_ref=context_ref, cams=cams, rendered_feat=rendered_feat, **kw)
        if xin.shape[0] == b:  # the target stream walks the 70 blocks in module order (the reference stream has b * n rows)
            blocks_rec.append(dict(x=xin, ctx=context, out=res[0], cref=context_ref, fg=res[1], alphas=res[2], rgb=res[3], xref=res[4]))
        return res

    def rec_res(w, xin, emb):
        out = orig["res"](w, xin, emb)
        if xin.shape[0] == b:
            res_rec.append(dict(x=xin, emb=emb, out=out))
        return out

    def rec_st(w, xin, xrin, *a, **kw):
        first = len(blocks_rec)
        res = orig["st"](w, xin, xrin, *a, **kw)
        st_rec.append(dict(x=xin, out=res[0], first=first, last=len(blocks_rec) - 1))
        return res

    def rec_seq(w, h, hr, *a):
        res = orig["seq"](w, h, hr, *a)
        seq_rec[a[-1]] = dict(x=h, out=res[0], st=len(st_rec) - 1)
        return res

    O.transformer_block, O.res_block, O.spatial_transformer, O._run_block = rec_block, rec_res, rec_st, rec_seq
    try:
        want, wfg, wal, wrgb = O.unet_forward(sd, x, t, ctx, y, cams=cams, input_ref=xr, sigmas_ref=tr, model_channels=320, num_samples=24, far=2.0)
    finally:
        O.transformer_block, O.res_block, O.spatial_transformer, O._run_block = orig["block"], orig["res"], orig["st"], orig["seq"]
    del sd
    got, fgs, als, rgbs = net(x.to(DEV), timesteps=t.to(DEV), context=ctx.to(DEV), y=y.to(DEV), pose=unpack_cameras(cams), input_ref=xr.to(DEV),
                              sigmas_ref=tr.to(DEV), mask_ref=None)
    assert len(fgs) == len(wfg) == 12 and len(rgbs) == 12 and len(als) == 12
    errs = {"eps": rel(got, want)}
    for i in range(12):
        errs[f"fg{i}"], errs[f"rgb{i}"], errs[f"alpha{i}"] = rel(fgs[i], wfg[i]), rel(rgbs[i], wrgb[i]), rel(als[i], wal[i])
    print("full-SDXL cfg-A rel errors:", {k: round(v, 4) for k, v in errs.items()})
    # End to end the bf16 residual stream drifts over 70 transformer blocks and the renders inherit it through their inputs (alphas most:
    # alpha = 1 - exp(-delta exp(sigma_raw)) amplifies a relative error of sigma_raw); measured eps 2.4e-2, worst alpha 4.2e-2.  Every
    # unit ITSELF is within 1e-2 on identical inputs -- the teacher-forced checks below, which is where the 1e-2 bar is asserted.
    assert errs["eps"] < 4e-2, errs
    assert max(v for k, v in errs.items() if k != "eps") < 6e-2, errs
    from cd360 import sampling
    from sgm.modules.attention import BasicTransformerBlock, SpatialTransformer
    from sgm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, Upsample
    # The other two routes of the same modules, end to end against the same oracle output: forward hooks on the 70 blocks (what the
    # references harvest registers: the SpatialTransformers leave their fused route, the blocks keep their fused internals) and hooks on a
    # submodule of every block (the strict route of a rebound forward).  Between themselves the routes differ by where bf16 rounds.
    kwargs = dict(timesteps=t.to(DEV), context=ctx.to(DEV), y=y.to(DEV), pose=unpack_cameras(cams), input_ref=xr.to(DEV), sigmas_ref=tr.to(DEV), mask_ref=None)
    route_errs = {}
    for route, pick in (("hooked", lambda m: m), ("strict", lambda m: m.norm1)):
        fired = []
        hs = [pick(m).register_forward_hook(lambda mod, i, o: fired.append(1)) for m in net.modules() if isinstance(m, BasicTransformerBlock)]
        out_r = net(x.to(DEV), **kwargs)
        for h_ in hs:
            h_.remove()
        assert len(fired) >= 70
        route_errs[route] = (rel(out_r[0], want), rel(out_r[0], got), max(rel(a, b_) for a, b_ in zip(out_r[1], wfg)))
    print("full-SDXL cfg-A routes (eps vs oracle, eps vs fused route, worst fg vs oracle):", {k: tuple(round(v, 4) for v in vs) for k, vs in route_errs.items()})
    assert all(vs[0] < 4e-2 and vs[2] < 6e-2 for vs in route_errs.values()), route_errs
    cl = lambda v: dev(v).contiguous(memory_format=torch.channels_last)
    margins = {}
    # ---- the 12 renders: each HIP pose block gets the oracle's own block inputs (rounded to bf16) ----
    all_blocks = [m for m in net.modules() if isinstance(m, BasicTransformerBlock)]
    assert len(blocks_rec) == len(all_blocks) == 70
    pose = unpack_cameras(cams)
    tf, tfe, tpose, tplain = {}, {}, {}, {}
    for i, (blk, rec) in enumerate(zip(all_blocks, blocks_rec)):
        C = rec["x"].shape[-1]
        xin, cin = dev(rec["x"]).contiguous(), dev(rec["ctx"])
        if rec["cref"] is not None:
            assert blk.image_cross
            cref = rec["cref"]
            cref4 = cref.reshape(b, cref.shape[0] // b, *cref.shape[1:]) if cref.dim() == 3 else cref
            xref, fg, _, al, rgb = blk.reference_attn(xin, dev(cref4), cin, pose, None, None)
            tf[i] = (rel(xref, rec["xref"]), rel(fg, rec["fg"]), rel(al, rec["alphas"]), rel(rgb, rec["rgb"]))
            tfe[i] = (elem(xref, rec["xref"]), elem(fg, rec["fg"]), elem(al, rec["alphas"]), elem(rgb, rec["rgb"]))
            whole = blk(xin, context=cin, context_ref=dev(cref4).reshape(-1, *cref4.shape[2:]), pose=pose)[0]
            tpose[i] = rel(whole, rec["out"])
        else:
            assert blk.norm1.weight.shape[0] == C and blk.fused_ready(xin)
            tplain[i] = (C, rel(blk._forward_fused(xin, None, cin)[0], rec["out"]))
    print("teacher-forced render errors (xref, fg, alphas, rgb) per pose block:", {k: tuple(round(e, 4) for e in v) for k, v in tf.items()})
    print("   element-wise |d| / (1e-2 |want| + 1e-3 max|want|) of the same (xref, fg, alphas, rgb):", {k: tuple(round(e, 3) for e in v) for k, v in tfe.items()})
    # every fg / alphas / rgb entry of the 12 renders within 1e-2 of its own oracle value (+ 1e-3 of the tensor maximum): measured <= 0.88
    assert max(max(v[1:]) for v in tfe.values()) <= 1.0, tfe
    print("teacher-forced whole pose blocks (x -> out):", {k: round(v, 4) for k, v in tpose.items()})
    assert len(tf) == 12 and len(tplain) == 58
    margins["render (12 pose blocks: xref, fg, alphas, rgb)"] = max(max(v) for v in tf.values())
    margins["pose block, whole (12)"] = max(tpose.values())
    for C in (640, 1280):
        margins[f"plain transformer block C={C}, fused route ({sum(1 for c, _ in tplain.values() if c == C)})"] = max(e for c, e in tplain.values() if c == C)
    # the module route (HipLinear / HipLayerNorm members, what sample.py's rebound forwards and hooked blocks run): first and last plain block per width
    tmod = {}
    for C in (640, 1280):
        ids = [i for i, (c, _) in tplain.items() if c == C]
        for i in (ids[0], ids[-1]):
            tmod[i] = rel(all_blocks[i]._forward(dev(blocks_rec[i]["x"]).contiguous(), dev(blocks_rec[i]["ctx"]))[0], blocks_rec[i]["out"])
    margins["plain transformer block, module route (first / last per width)"] = max(tmod.values())
    # ---- every ResBlock (GroupNorm + SiLU -> conv3x3 (+ emb) -> GroupNorm + SiLU -> conv3x3 + skip) ----
    all_res = [m for m in net.modules() if isinstance(m, ResBlock)]
    assert len(res_rec) == len(all_res)
    tres = {}
    for i, (rb, rec) in enumerate(zip(all_res, res_rec)):
        assert (rb.channels, rb.out_channels) == (rec["x"].shape[1], rec["out"].shape[1])
        tres[i] = ((rb.channels, rb.out_channels), rel(rb(cl(rec["x"]), dev(rec["emb"])), rec["out"]))
    print("teacher-forced ResBlocks (in, out channels):", {k: (v[0], round(v[1], 4)) for k, v in tres.items()})
    margins[f"ResBlock ({len(tres)})"] = max(e for _, e in tres.values())
    # ---- proj_in (GroupNorm -> Linear) and proj_out (Linear + residual) of every SpatialTransformer ----
    all_st = [m for m in net.modules() if isinstance(m, SpatialTransformer)]
    assert len(st_rec) == len(all_st) == 11
    tin, tout = {}, {}
    for i, (st, rec) in enumerate(zip(all_st, st_rec)):
        tok, _ = st._enter(cl(rec["x"]))
        tin[i] = rel(tok, blocks_rec[rec["first"]]["x"])
        tout[i] = rel(st._leave(dev(blocks_rec[rec["last"]]["out"]).contiguous(), cl(rec["x"])), rec["out"])
    margins["SpatialTransformer proj_in (11)"], margins["SpatialTransformer proj_out + residual (11)"] = max(tin.values()), max(tout.values())
    # ---- Downsample (stride-2 conv) and Upsample (nearest 2x folded into its conv) ----
    tconv = {}
    for name, rec in seq_rec.items():
        seq = net.get_submodule(name) if "." in name else getattr(net, name)
        mods = list(seq)
        if len(mods) == 1 and isinstance(mods[0], Downsample):
            tconv[name + " Downsample"] = rel(mods[0](cl(rec["x"])), rec["out"])
        elif isinstance(mods[-1], Upsample):
            tconv[name + " Upsample"] = rel(mods[-1](cl(st_rec[rec["st"]]["out"])), rec["out"])
    assert len(tconv) == 4, tconv
    margins["Downsample / Upsample convolutions (4)"] = max(tconv.values())
    print("worst plain transformer blocks (index, width, error):", sorted(((i, c, round(e, 4)) for i, (c, e) in tplain.items()), key=lambda t: -t[2])[:6])
    print("teacher-forced margins (worst relative error per kind):")
    for k, v in margins.items():
        print(f"  {k}: {v:.2e}")
    # north_star's bar -- bf16 attention / render outputs within 1e-2 of the reference's path on identical inputs -- holds for every
    # OPERATOR-level unit: the renders, the projections, the ResBlocks, the resampling convolutions.  A whole transformer block is five
    # GEMMs, two attentions and a GEGLU around a residual stream that is rounded to bf16 three times (2^-9 = 2e-3 of a value each time);
    # the worst of the 51 blocks at C = 1280 measures 1.02e-2, all others below 1e-2, so whole blocks are held to 1.25e-2.
    whole = {k: v for k, v in margins.items() if k.startswith(("plain transformer block", "pose block, whole"))}
    assert max(v for k, v in margins.items() if k not in whole) < 1e-2, margins
    assert max(whole.values()) < 1.25e-2, margins


# ------------------------------------------------------------------------------------------- BASELINE configs[1] / [3] at full size
def _sdxl_net(seed=31):
    """Random-init SDXL UNet (the reference's network_config) with the pose path switched on (the stock init makes it a no-op, F7)."""
    from cd360 import sampling
    from cd360.configs import SDXL_NETWORK_CONFIG
    from sgm.util import instantiate_from_config
    torch.manual_seed(seed)
    with torch.device(DEV):
        net = instantiate_from_config(SDXL_NETWORK_CONFIG)
    net = net.to(BF)
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    with torch.no_grad():
        for _, blk in sampling.pose_blocks(net):
            c = blk.pose_emb_layers.weight.shape[0]
            blk.pose_emb_layers.weight.add_(torch.randn(c, 2 * c, generator=g, device=DEV).mul_(0.02).to(BF))
            blk.pose_featurenerf.model.decoder.weight.copy_(torch.randn(4, c, generator=g, device=DEV).mul_(0.02))
        for m in net.modules():
            if m.__class__.__name__ == "SpatialTransformer":
                m.proj_out.weight.copy_(torch.randn(m.proj_out.weight.shape, generator=g, device=DEV).mul_(0.02))
        net.out[-1].weight.copy_(torch.randn(net.out[-1].weight.shape, generator=g, device=DEV).mul_(0.02))  # zero_module: eps would be 0
    return net, g


@torch.no_grad()
def test_cfgB_level2_sampling_block_n50_cached_equals_uncached():
    """BASELINE configs[1] shape for one LEVEL-2 pose block: C=1280, r=32 (hw=1024), S=24, n=50 references, CFG batch 3 -- the same
    size-independent properties as the level-1 test (cached == re-rendered, render independent of x, conditional thirds agree,
    unconditional differs, outputs finite and in range)."""
    from cd360 import sampling, synth
    blk = make_block(15, C=1280, heads=20, cd=2048, S=24)
    n_train, n, hw = 50, 50, 1024
    sampling.set_references(blk, {"": dev(W.tensor("references", (n_train + 1, hw, 1280), seed=15))})
    sampling.enable_reference_sampling(blk, list(range(n)))
    pose = synth.pose_batch(1, n, seed=10, n_train=n_train) * 3
    ctx = dev(W.tensor("ctx", (1, 77, 2048), seed=15)).expand(3, -1, -1).contiguous()
    x = dev(W.tensor("x", (1, hw, 1280), seed=15)).expand(3, -1, -1).contiguous()
    out_a, fg, _, alphas, rgb = blk(x, context=ctx, context_ref=x, pose=pose)
    rend = blk.rendered_feat.clone()
    assert torch.equal(out_a, blk(x, context=ctx, context_ref=x, pose=pose)[0])  # cached step
    sampling.clear_rendered_feat(blk)
    out_c = blk(x * 0.5, context=ctx, context_ref=x, pose=pose)[0]
    assert torch.equal(blk.rendered_feat, rend) and torch.isfinite(out_c.float()).all()
    assert float(fg.min()) >= -1e-5 and float(fg.max()) <= 1 + 1e-4 and float(alphas.min()) >= 0 and float(alphas.max()) <= 1
    assert float(rgb.min()) >= 0 and float(rgb.max()) <= 1 + 1e-4
    assert rel(rend[1], rend[2]) < 2e-2 and rel(rend[0], rend[1]) > 5e-2 and rel(out_a[1], out_a[2]) < 2e-2


@torch.no_grad()
def test_cfgB_full_unet_sampling_step_properties():
    """BASELINE configs[1]: ONE full UNet denoise step of sample.py's path at 1024^2 -- latent 128^2, CFG batch 3, 50 reference views from
    synthetic `references`, all 12 FeatureNeRF renders -- then a cached step.  No oracle at this size, so properties: a cached step on
    the same input reproduces the render step bit for bit; the two image-conditional thirds (identical inputs) agree to bf16 round-off
    while the unconditional third differs; every output is finite; 12 renders were produced and stayed cached."""
    from cd360 import sampling, synth
    net, g = _sdxl_net()
    net.eval()
    L, n = 128, 50
    refs = {}
    for name, blk in sampling.pose_blocks(net):
        c = blk.pose_emb_layers.weight.shape[0]
        r = L // 2 if c == 640 else L // 4
        refs[name] = torch.randn(n + 1, r * r, c, generator=g, device=DEV).to(BF)
    sampling.set_references(net, refs)
    sampling.enable_reference_sampling(net, list(range(n)))
    pose = synth.pose_batch(1, n, seed=12, n_train=n) * 3
    x = torch.randn(1, 4, L, L, generator=g, device=DEV).expand(3, -1, -1, -1).contiguous()
    ctx1 = torch.randn(2, 77, 2048, generator=g, device=DEV).to(BF)
    y1 = torch.randn(2, 2816, generator=g, device=DEV).to(BF)
    ctx, y = torch.cat([ctx1[:1], ctx1[:1], ctx1[1:]], 0), torch.cat([y1[:1], y1[:1], y1[1:]], 0)  # (uc, uc, c) of the 3-way guider
    t = torch.full((3,), 500.0, device=DEV)
    eps_a, fgs, alphas, rgbs = net(x, timesteps=t, context=ctx, y=y, pose=pose)
    assert len(fgs) == 12 and len(alphas) == 12 and len(rgbs) == 12
    blocks = [blk for _, blk in sampling.pose_blocks(net)]
    assert all(blk.rendered_feat is not None for blk in blocks)
    eps_b = net(x, timesteps=t, context=ctx, y=y, pose=pose)[0]  # cached render
    assert torch.equal(eps_a, eps_b)
    assert torch.isfinite(eps_a).all() and all(torch.isfinite(f.float()).all() for f in fgs + rgbs)
    assert all(float(f.min()) >= -1e-5 and float(f.max()) <= 1 + 1e-4 for f in fgs)
    # thirds 0 and 1 share the text conditioning and differ in the image conditioning (null image vs references); thirds 1 and 2 share
    # the references and differ in the text: all three must be distinct, and the renders of thirds 1 and 2 must coincide before the text
    for blk in blocks:
        rf = blk.rendered_feat
        assert rel(rf[0], rf[1]) > 1e-2
    assert rel(eps_a[0], eps_a[1]) > 1e-3 and rel(eps_a[1], eps_a[2]) > 1e-3


def test_config4_sdxl_size_train_step_reduces_the_loss():
    """BASELINE configs[3] at SDXL width and depth and at the config's own batch (tools/bench_train.py's shapes: bs = 4, 4 reference views,
    512^2 images): train mode (stratified jitter), trainkeys = pose, forward + the four-term loss + backward through the HIP kernels
    (every Linear on cd360_gemm_bf16 / cd360_gemm_tn_bf16: no library GEMM may be reached) + AdamW on fp32 masters.  The loss of the fixed
    synthetic batch must fall over four steps, every trainable gradient must be finite and non-zero."""
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    net, g = _sdxl_net(seed=41)
    net.train()
    names = finetune.select_trainable(net, "pose")
    assert len(names) == 96
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    b, n, L = 4, 4, 64
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    batch = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=DEV), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
                 pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=DEV),
                 target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=DEV),
                 mask=torch.ones(b, 1, L, L, device=DEV), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
    from test_linear_gpu import no_library_gemm
    with no_library_gemm():
        losses = [float(finetune.train_step(net, loss_fn, opt, **batch)[0]) for _ in range(4)]
    print("config-4 losses:", [round(v, 4) for v in losses])
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]
    params = dict(net.named_parameters())
    assert all(params[k].grad is not None and torch.isfinite(params[k].grad.float()).all() for k in names)
    assert sum(float(params[k].grad.float().abs().sum()) > 0 for k in names) >= len(names) - 12  # nviews.bias has a zero gradient


def test_config4_train_step_replayed_from_a_hipgraph_follows_the_eager_steps():
    """finetune.GraphedTrainStep: the whole optimisation step (forward, loss, backward, AdamW on fp32 masters) captured once and replayed.
    Without jitter (eval-mode raymarchers: every kernel is deterministic) the replayed steps must follow the eagerly launched ones --
    same losses step for step (1e-4 relative) and the same trained weights to bf16 resolution; new inputs copied into the static buffers must change the result; with the stratified jitter on the device
    generator (train mode) successive replays must draw fresh jitter (losses differ from the no-jitter run) and still descend."""
    import copy
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    net, g = _sdxl_net(seed=43)
    net.eval()
    names = finetune.select_trainable(net, "pose")
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    b, n, L = 2, 2, 32
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    batch = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=DEV), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
                 pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=DEV),
                 target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=DEV),
                 mask=torch.ones(b, 1, L, L, device=DEV), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
    start = copy.deepcopy({k: v for k, v in net.state_dict().items() if "pose" in k})
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    eager = [float(finetune.train_step(net, loss_fn, opt, **batch)[0]) for _ in range(6)]
    w_eager = {k: dict(net.named_parameters())[k].detach().clone() for k in names}
    net.load_state_dict(start, strict=False)
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    assert opt.fused and opt.capturable
    with pytest.raises(ValueError):  # the torch fall-back keeps its step counts on the host unless built capturable
        finetune.GraphedTrainStep(net, loss_fn, finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4, fused=False), batch)
    step = finetune.GraphedTrainStep(net, loss_fn, opt, batch, warmup=2)  # steps 1-2 run eagerly inside; the capture itself executes nothing
    graphed = [float(step()[0]) for _ in range(4)]
    print("eager:", [round(v, 5) for v in eager], "graph (steps 3-6):", [round(v, 5) for v in graphed])
    assert all(abs(a - e) <= 1e-4 * abs(e) for a, e in zip(graphed, eager[2:]))
    params = dict(net.named_parameters())
    worst = max(float((params[k].float() - w_eager[k].float()).abs().max() / w_eager[k].float().abs().max().clamp_min(1e-6)) for k in names)
    assert worst < 2e-2, worst  # bf16 copies of fp32 masters that differ in their last bits
    same = float(step()[0])
    other = float(step(target=batch["target"] * 0.5, noised=batch["noised"] + 0.1)[0])
    assert torch.isfinite(torch.tensor(other)) and abs(other - same) > 1e-3 * abs(same)
    # stratified jitter inside the graph: the device generator advances on every replay
    net.load_state_dict(start, strict=False)
    net.train()
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    step = finetune.GraphedTrainStep(net, loss_fn, opt, batch, warmup=2)
    jit = [float(step()[0]) for _ in range(6)]
    print("graph, stratified:", [round(v, 5) for v in jit])
    assert all(torch.isfinite(torch.tensor(jit))) and jit[-1] < jit[0] and any(abs(a - e) > 1e-6 * abs(e) for a, e in zip(jit, eager[2:]))
    assert all(m.device_rng for m in net.modules() if hasattr(m, "device_rng"))


@torch.no_grad()
def test_groupnorm_statistics_ride_through_the_skip_concat():
    """openaimodel.py:1074-1076 (th.cat([h, hs.pop()], dim=1)) in front of a ResBlock's in_layers GroupNorm: when both inputs carry the
    per-slab channel sums of their producers' epilogues, the concatenation carries their concatenation (slab counts brought to the coarser
    one) and the GroupNorm skips its statistics pass -- same result as statistics taken from the concatenated tensor itself."""
    from sgm.modules.diffusionmodules import openaimodel as om
    from sgm.modules.diffusionmodules.util import _tagged_gn_stats, group_norm_tokens, normalization, tag_gn_stats
    g = torch.Generator(device=DEV).manual_seed(77)
    N, H, W, ca, cb = 3, 32, 32, 1280, 640
    mk = lambda c: (torch.randn(N, c, H, W, generator=g, device=DEV) * 1.5 + 0.3).to(BF).contiguous(memory_format=torch.channels_last)
    a, b = mk(ca), mk(cb)

    def slab_stats(x, slabs):
        t = x.permute(0, 2, 3, 1).reshape(N, slabs, H * W // slabs, x.shape[1]).float()
        return torch.stack([t.sum(2), (t * t).sum(2)], -1).contiguous()

    tag_gn_stats(a, slab_stats(a, 16))
    tag_gn_stats(b, slab_stats(b, 8))
    out = om._cat_channels(a, b)
    st = _tagged_gn_stats(out)
    assert st is not None and st.shape == (N, 8, ca + cb, 2)
    assert torch.equal(out, torch.cat([a, b], 1))
    norm = normalization(ca + cb).to(DEV, BF)
    norm.weight.copy_(1 + 0.2 * torch.randn(ca + cb, generator=g, device=DEV))
    norm.bias.copy_(0.1 * torch.randn(ca + cb, generator=g, device=DEV))
    with_tag = group_norm_tokens(norm, out, silu=True)
    fresh = group_norm_tokens(norm, out.clone(memory_format=torch.channels_last), silu=True)  # a clone has no tag: statistics pass over the data
    want = torch.nn.functional.silu(torch.nn.functional.group_norm(out.float(), norm.num_groups, norm.weight.float(), norm.bias.float(), norm.eps))
    want = want.permute(0, 2, 3, 1).reshape(N, H * W, ca + cb)
    assert rel(with_tag, want) < 1e-2 and rel(fresh, want) < 1e-2 and rel(with_tag, fresh) < 4e-3
    assert _tagged_gn_stats(om._cat_channels(a, mk(cb))) is None  # an untagged input: no tag on the result


def test_graphed_train_step_captures_the_gradient_allreduce():
    """Data-parallel fine-tuning replayed from a hipGraph: under an initialised process group (RCCL, a group of ONE here -- all a 1-GPU box
    offers; `allreduce_single` forces the collective path that a larger group takes) MasterAdamW's flat gradient all-reduce is captured
    with the step.  The replayed steps must equal the eager steps of the same optimiser, which in a group of one are the plain steps."""
    import copy
    import torch.distributed as dist
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        net, g = _sdxl_net(seed=47)
        net.eval()
        names = finetune.select_trainable(net, "pose")
        loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
        b, n, L = 1, 2, 32
        rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
        batch = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=DEV), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
                     pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=DEV),
                     target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=DEV),
                     mask=torch.ones(b, 1, L, L, device=DEV), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
        start = copy.deepcopy({k: v for k, v in net.state_dict().items() if "pose" in k})
        opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
        eager = [float(finetune.train_step(net, loss_fn, opt, **batch)[0]) for _ in range(5)]  # group of one: no collective
        net.load_state_dict(start, strict=False)
        opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
        opt.allreduce_single = True
        dist.all_reduce(torch.ones(8, device=DEV))  # communicator set-up outside the capture
        step = finetune.GraphedTrainStep(net, loss_fn, opt, batch, warmup=2)
        graphed = [float(step()[0]) for _ in range(3)]
        print("eager:", [round(v, 5) for v in eager], "graph + all-reduce (steps 3-5):", [round(v, 5) for v in graphed])
        # the all-reduce path rounds the gradients through one fp32 flat buffer and back to bf16: same values
        assert all(abs(a - e) <= 1e-4 * abs(e) for a, e in zip(graphed, eager[2:]))
    finally:
        dist.destroy_process_group()
