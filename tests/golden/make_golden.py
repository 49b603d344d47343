"""Generate golden vectors by running the reference's own modules (CPU, fp32) in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes tests/golden/*.npz (+ state_dict key/shape listings as .json.gz).  The fixtures hold data
only: inputs that cannot be regenerated and the outputs the reference produced.  Weights are
regenerated on both sides by tests/golden/weights.py.  Cameras come from cd360.synth
(deterministic).  The reference is imported through tests/golden/refshim.py; its third-party
dependencies (pytorch3d / xformers / omegaconf) are stand-ins, so these vectors pin the
reference's *own* code, not those libraries ("parity unpinned" at that boundary).
"""
from __future__ import annotations

import ast
import gzip
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402
import weights as W  # noqa: E402

ns = refshim.import_reference()
from cd360 import synth  # noqa: E402
from cd360.cameras import pack_cameras  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


class Recorder:
    """Records grid_sample calls and the random draws made inside a reference forward."""

    def __init__(self):
        self.grid_calls, self.rand = [], []

    def __enter__(self):
        F = torch.nn.functional
        self._gs, self._rand, self._rand_like = F.grid_sample, torch.rand, torch.rand_like

        def gs(inp, grid, **kw):
            out = self._gs(inp, grid, **kw)
            self.grid_calls.append((inp.clone(), grid.clone(), out.clone()))
            return out

        def rand(*a, **k):
            t = self._rand(*a, **k)
            self.rand.append(t.clone())
            return t

        def rand_like(x, **k):
            t = self._rand_like(x, **k)
            self.rand.append(t.clone())
            return t

        F.grid_sample, torch.rand, torch.rand_like = gs, rand, rand_like
        return self

    def __exit__(self, *a):
        F = torch.nn.functional
        F.grid_sample, torch.rand, torch.rand_like = self._gs, self._rand, self._rand_like


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if v is None:
            continue
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def keyfile(name, module):
    sd = module.state_dict()
    listing = {k: list(v.shape) for k, v in sd.items()}
    with gzip.open(os.path.join(HERE, name + ".keys.json.gz"), "wt") as f:
        json.dump(listing, f)
    return listing


def jitter_kw(rec):
    """rand draws in call order: xs jitter [r+1], ys jitter [r+1], depth jitter [hw,S+1] (train mode)."""
    if not rec.rand:
        return {}
    return dict(jit_x=rec.rand[0], jit_y=rec.rand[1], jit_d=rec.rand[2])


# ---------------------------------------------------------------- 1. NerfSDModule
def case_nerf(train: bool):
    C, r, n, S, b = 64, 8, 2, 4, 2
    m = ns.nerf.NerfSDModule(mode="feature-nerf", out_channels=C, far_plane=2.0, num_samples=S, rgb_predict=True, stratified=True)
    W.load_into(m, seed=1)
    m.train(train)
    pose = synth.pose_batch(b, n, seed=3)
    xref = W.tensor("xref", (b, n, r * r, C), seed=1)
    torch.manual_seed(11)
    ray_out = {}
    h = m.raymarcher.register_forward_hook(lambda mod, i, o: ray_out.update(rays=o[0], points=o[1], dists=o[2]))
    with Recorder() as rec:
        feats, sigma, dists, attn, rgb, _, _ = m(pose, xref)
    h.remove()
    _, grid, plane = rec.grid_calls[0]
    npz("nerf_train" if train else "nerf_eval", cams=pack_cameras(pose), xref=xref, rays=ray_out["rays"], points=ray_out["points"],
        dists=dists, grid=grid, plane=plane, feats=feats, sigma=sigma, view_weights=attn, rgb=rgb, **jitter_kw(rec))
    return m


# ---------------------------------------------------------------- 2. BasicTransformerBlock with pose
def make_block(C, heads, ctx_dim, S):
    return ns.attention.BasicTransformerBlock(C, heads, 64, context_dim=ctx_dim, checkpoint=False, attn_mode="softmax-xformers",
                                              image_cross=True, far=2, num_samples=S, rgb_predict=True, mode="feature-nerf", stratified=True)


def case_block(train: bool):
    C, heads, r, n, S, b, T, cd = 64, 1, 8, 2, 4, 2, 77, 32
    blk = make_block(C, heads, cd, S)
    W.load_into(blk, seed=2)
    blk.train(train)
    pose = synth.pose_batch(b, n, seed=4)
    x = W.tensor("x", (b, r * r, C), seed=2)
    ctx = W.tensor("ctx", (b, T, cd), seed=2)
    cref = W.tensor("cref", (b * n, r * r, C), seed=2)
    torch.manual_seed(12)
    with Recorder() as rec:
        out, fg, wts, alphas, rgb = blk(x, context=ctx, context_ref=cref, pose=pose)
    assert wts is None
    plain = blk(x, context=ctx)[0]
    npz("block_train" if train else "block_eval", cams=pack_cameras(pose), x=x, ctx=ctx, cref=cref, out=out, fg=fg, alphas=alphas, rgb=rgb,
        plain=plain, **jitter_kw(rec))
    if not train:
        keyfile("block", blk)


from make_golden_params import sdxl_block_inputs  # noqa: E402  (shared with the tests: inputs regenerated on both sides)


def case_block_sdxl():
    """"plus one SDXL-dim block slice".  The pose block at the two widths of the shipped config
    (configs/train_co3d_concept.yaml:27-54: C = 640 with 10 heads, C = 1280 with 20 heads, head dim 64, text context 2048 wide, 77
    tokens) on a small ray grid (r = 8, n = 2 reference views, S = 4 samples), eval mode: the reference's own BasicTransformerBlock
    (attention.py:428-637) end to end -- FeatureNeRF render, pose-token cross-attention over the text context, volume render, injection,
    GEGLU feed-forward -- plus the same block called without a pose."""
    out = {}
    for C, heads in ((640, 10), (1280, 20)):
        blk = make_block(C, heads, 2048, 4)
        W.load_into(blk, seed=6)
        blk.eval()
        x, ctx, cref, pose = sdxl_block_inputs(C)
        o, fg, wts, alphas, rgb = blk(x, context=ctx, context_ref=cref, pose=pose)
        assert wts is None
        out.update({f"c{C}_cams": pack_cameras(pose), f"c{C}_out": o, f"c{C}_fg": fg, f"c{C}_alphas": alphas, f"c{C}_rgb": rgb,
                    f"c{C}_plain": blk(x, context=ctx)[0]})
    npz("block_sdxl", **out)


# ---------------------------------------------------------------- 2b. mask_ref (nerfsd_pytorch3d.py:61-70; live in config 4: data_co3d.py:485, loss.py:154)
def case_mask_ref():
    """The reference-view masks: NerfSDModule, the pose block (eval, and train mode with the jitter draws recorded plus the reference's
    own autograd gradients of the 'pose' parameters) and the tiny UNet, all with a [b, n, 1, Hm, Wm] 0/1 mask that the module
    nearest-resizes to the feature-map side and multiplies onto the reference features."""
    out = {}
    # NerfSDModule (same weights / inputs as case_nerf)
    C, r, n, S, b = 64, 8, 2, 4, 2
    m = ns.nerf.NerfSDModule(mode="feature-nerf", out_channels=C, far_plane=2.0, num_samples=S, rgb_predict=True, stratified=True)
    W.load_into(m, seed=1)
    m.eval()
    pose = synth.pose_batch(b, n, seed=3)
    xref = W.tensor("xref", (b, n, r * r, C), seed=1)
    mask = (W.tensor("mask_ref", (b, n, 1, 4 * r, 4 * r), seed=1) > -0.3).float()  # ~60 % ones, 4x the feature-map side
    feats, sigma, dists, attn, rgb, _, _ = m(pose, xref, mask_ref=mask)
    out.update(nerf_cams=pack_cameras(pose), nerf_xref=xref, nerf_mask=mask, nerf_feats=feats, nerf_sigma=sigma, nerf_view_weights=attn, nerf_rgb=rgb)
    # pose block, eval and train (same weights / inputs as case_block)
    C, heads, r, n, S, b, T, cd = 64, 1, 8, 2, 4, 2, 77, 32
    pose = synth.pose_batch(b, n, seed=4)
    x = W.tensor("x", (b, r * r, C), seed=2)
    ctx = W.tensor("ctx", (b, T, cd), seed=2)
    cref = W.tensor("cref", (b * n, r * r, C), seed=2)
    mask = (W.tensor("mask_ref", (b, n, 1, 2 * r, 2 * r), seed=2) > -0.3).float()
    out.update(blk_cams=pack_cameras(pose), blk_x=x, blk_ctx=ctx, blk_cref=cref, blk_mask=mask)
    for train in (False, True):
        blk = make_block(C, heads, cd, S)
        W.load_into(blk, seed=2)
        blk.train(train)
        tag = "train" if train else "eval"
        torch.manual_seed(12)
        for name, p_ in blk.named_parameters():
            p_.requires_grad = train and "pose" in name
        with Recorder() as rec, torch.set_grad_enabled(train):
            o, fg, wts, alphas, rgb = blk(x, context=ctx, context_ref=cref, pose=pose, mask_ref=mask)
            if train:
                cot = W.tensor("cot", tuple(o.shape), seed=2)
                ((o * cot).sum() + fg.sum() + rgb.sum()).backward()
        out.update({f"blk_{tag}_out": o.detach(), f"blk_{tag}_fg": fg.detach(), f"blk_{tag}_alphas": alphas.detach(), f"blk_{tag}_rgb": rgb.detach()})
        if train:
            out.update({f"blk_train_{k}": v for k, v in jitter_kw(rec).items()})
            out.update({f"blk_grad.{name}": p_.grad for name, p_ in blk.named_parameters() if p_.requires_grad and p_.grad is not None})
    # tiny UNet (same weights / inputs as case_unet)
    net = ns.openaimodel.UNetModel(**UNET_TINY)
    W.load_into(net, seed=5)
    net.eval()
    b, n, L, T = 1, 2, 16, 77
    pose = synth.pose_batch(b, n, seed=7)
    x = W.tensor("x", (b, 4, L, L), seed=5)
    xin = W.tensor("input_ref", (b, n, 4, L, L), seed=5)
    ctx = W.tensor("ctx", (b + b * n, T, 32), seed=5)
    y = W.tensor("y", (b + b * n, 16), seed=5)
    mask = (W.tensor("mask_ref", (b, n, 1, 8 * L, 8 * L), seed=5) > -0.3).float()  # image-resolution mask (data_co3d.py:485)
    eps, fgs, alphas, rgbs = net(x, timesteps=torch.tensor([500.0]), context=ctx, y=y, pose=pose, input_ref=xin, sigmas_ref=torch.tensor([120.0]),
                                 mask_ref=mask)
    out.update(unet_mask=mask, unet_out=eps, **{f"unet_fg{i}": v for i, v in enumerate(fgs)}, **{f"unet_rgb{i}": v for i, v in enumerate(rgbs)})
    npz("mask_ref", **out)


# ---------------------------------------------------------------- 3. SpatialTransformer dual stream
def make_st(C, heads, depth, cd, S):
    return ns.attention.SpatialTransformer(C, heads, 64, depth=depth, context_dim=cd, use_linear=True, attn_type="softmax-xformers",
                                           use_checkpoint=False, image_cross=True, rgb_predict=True, far=2, num_samples=S,
                                           mode="feature-nerf", stratified=True)


def case_st_dual():
    C, heads, depth, r, n, S, b, T, cd = 128, 2, 5, 8, 2, 4, 1, 77, 32
    st = make_st(C, heads, depth, cd, S)
    W.load_into(st, seed=3)
    st.eval()
    pose = synth.pose_batch(b, n, seed=5)
    x = W.tensor("x", (b, C, r, r), seed=3)
    xr = W.tensor("xr", (b * n, C, r, r), seed=3)
    ctx = W.tensor("ctx", (b, T, cd), seed=3)
    ctxr = W.tensor("ctxr", (b * n, T, cd), seed=3)
    out, xro, fgs, pw, alphas, rgbs = st(x, xr, context=ctx, contextr=ctxr, pose=pose)
    assert pw is None and len(fgs) == 2
    plain = st(x, None, context=ctx)[0]
    npz("st_dual", cams=pack_cameras(pose), x=x, xr=xr, ctx=ctx, ctxr=ctxr, out=out, xr_out=xro, fg0=fgs[0], fg1=fgs[1],
        alphas0=alphas[0], alphas1=alphas[1], rgb0=rgbs[0], rgb1=rgbs[1], plain=plain)
    keyfile("st", st)


# ---------------------------------------------------------------- 4. sample.py's patched forwards (cached render, CFG x3)
def _sample_py_functions():
    src = open(os.path.join(refshim.REF_ROOT, "sample.py")).read()
    tree = ast.parse(src)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("customforward", "_customforward")]
    # fingerprints of the two functions (a hash of the normalised AST: data about the reference, not its text) -- what
    # cd360/sample_py_patch.py compares a rebound forward with on the user's machine
    from cd360 import sample_py_patch
    import json
    with open(os.path.join(HERE, "sample_py_fingerprints.json"), "w") as f:
        json.dump({n.name: sample_py_patch.fingerprint_node(n) for n in wanted}, f, indent=1, sort_keys=True)
        f.write("\n")
    mod = ast.Module(body=wanted, type_ignores=[])
    from einops import rearrange
    env = {"torch": torch, "rearrange": rearrange, "choices": []}
    exec(compile(mod, "sample.py", "exec"), env)
    return env


def case_customforward():
    C, heads, depth, r, S, T, cd, n_train = 128, 2, 5, 8, 4, 77, 32, 4
    st = make_st(C, heads, depth, cd, S)
    W.load_into(st, seed=4)
    st.eval()
    env = _sample_py_functions()
    env["choices"][:] = [0, 2]  # global `choices` (sample.py:274-277)
    st.forward = env["customforward"].__get__(st, st.__class__)
    refs = {}
    for d, blk in enumerate(st.transformer_blocks):
        blk.forward = env["_customforward"].__get__(blk, blk.__class__)
        if hasattr(blk, "pose_emb_layers"):
            refs[d] = W.tensor(f"references.{d}", (n_train + 1, r * r, C), seed=4)
            blk.register_buffer("references", refs[d])
    pose1 = synth.pose_batch(1, 2, seed=6, n_train=n_train)
    pose = pose1 * 3  # 3-way CFG (sample.py:166-171)
    x0 = W.tensor("x0", (3, C, r, r), seed=4)
    x1 = W.tensor("x1", (3, C, r, r), seed=4)
    ctx = W.tensor("ctx", (3, T, cd), seed=4)
    out0, _, fgs, _, alphas, rgbs = st(x0, None, context=ctx, pose=pose)
    rend = {d: blk.rendered_feat.clone() for d, blk in enumerate(st.transformer_blocks) if getattr(blk, "rendered_feat", None) is not None}
    out1 = st(x1, None, context=ctx, pose=pose)[0]
    npz("customforward_cfg3", cams=pack_cameras(pose), x0=x0, x1=x1, ctx=ctx, out0=out0, out1=out1, fg0=fgs[0], fg1=fgs[1],
        rgb0=rgbs[0], rgb1=rgbs[1], rend0=rend[0], rend4=rend[4], choices=np.array([0, 2]))


# ---------------------------------------------------------------- 5. UNet (reduced depth/width), dual-stream eval
from make_golden_params import UNET_TINY  # noqa: E402


def case_unet():
    net = ns.openaimodel.UNetModel(**UNET_TINY)
    W.load_into(net, seed=5)
    net.eval()
    b, n, L, T = 1, 2, 16, 77
    pose = synth.pose_batch(b, n, seed=7)
    x = W.tensor("x", (b, 4, L, L), seed=5)
    xin = W.tensor("input_ref", (b, n, 4, L, L), seed=5)
    ctx = W.tensor("ctx", (b + b * n, T, 32), seed=5)
    y = W.tensor("y", (b + b * n, 16), seed=5)
    t = torch.tensor([500.0])
    sref = torch.tensor([120.0])
    out, fgs, alphas, rgbs = net(x, timesteps=t, context=ctx, y=y, pose=pose, input_ref=xin, sigmas_ref=sref, mask_ref=None)
    assert len(fgs) == 3
    npz("unet_tiny", cams=pack_cameras(pose), x=x, input_ref=xin, ctx=ctx, y=y, t=t, sigmas_ref=sref, out=out,
        **{f"fg{i}": v for i, v in enumerate(fgs)}, **{f"alphas{i}": v for i, v in enumerate(alphas)}, **{f"rgb{i}": v for i, v in enumerate(rgbs)})
    keyfile("unet_tiny", net)


def unet_grad_loss(out, fgs, rgbs):
    """The scalar both sides differentiate: fixed random cotangents on eps, every fg mask and every predicted rgb."""
    loss = (out.float() * W.tensor("g_out", tuple(out.shape), seed=5).to(out.device)).sum()
    for i, (fg, rgb) in enumerate(zip(fgs, rgbs)):
        loss = loss + (fg.float() * W.tensor(f"g_fg{i}", tuple(fg.shape), seed=5).to(fg.device)).sum()
        loss = loss + (rgb.float() * W.tensor(f"g_rgb{i}", tuple(rgb.shape), seed=5).to(rgb.device)).sum()
    return loss


def case_unet_grads():
    """Gradients of the trainable ('pose' in the name, diffusion.py:139-144) parameters of the tiny UNet, by the reference's own
    autograd (eval mode: no jitter; same inputs as unet_tiny.npz).  Pins the backward of the whole path: the data gradient runs
    from eps back through every layer downstream of the first pose block."""
    net = ns.openaimodel.UNetModel(**UNET_TINY)
    W.load_into(net, seed=5)
    net.eval()
    for name, p_ in net.named_parameters():
        p_.requires_grad = "pose" in name
    b, n, L, T = 1, 2, 16, 77
    pose = synth.pose_batch(b, n, seed=7)
    x = W.tensor("x", (b, 4, L, L), seed=5)
    xin = W.tensor("input_ref", (b, n, 4, L, L), seed=5)
    ctx = W.tensor("ctx", (b + b * n, T, 32), seed=5)
    y = W.tensor("y", (b + b * n, 16), seed=5)
    with torch.enable_grad():
        out, fgs, alphas, rgbs = net(x, timesteps=torch.tensor([500.0]), context=ctx, y=y, pose=pose, input_ref=xin,
                                     sigmas_ref=torch.tensor([120.0]), mask_ref=None)
        unet_grad_loss(out, fgs, rgbs).backward()
    grads = {name: p_.grad for name, p_ in net.named_parameters() if p_.requires_grad}
    assert all(g is not None for g in grads.values()), [k for k, g in grads.items() if g is None]
    npz("unet_tiny_grads", **grads)


def dummy_network(x_in, c_noise, cond, **kw):
    """Deterministic stand-in for OpenAIWrapper(UNet): depends on every input the sampler stack prepares."""
    b = x_in.shape[0]
    ctx = cond["crossattn"][:b].float().mean((1, 2)).view(-1, 1, 1, 1)
    vec = cond["vector"][:b].float().mean(1).view(-1, 1, 1, 1)
    pred = 0.3 * torch.tanh(x_in) + 0.001 * c_noise.float().view(-1, 1, 1, 1) / 10 + 0.1 * ctx + 0.05 * vec
    return pred, [], [], [torch.zeros(b, 4, 3)]


def case_sampler():
    """EulerEDMSampler + ScheduledCFGImgTextRef + DiscreteDenoiser(EpsScaling, LegacyDDPM) exactly as sample.py configures them
    (sample.py:230-240, configs/train_co3d_concept.yaml:14-25,119-131), around a deterministic dummy network."""
    den = ns.denoiser.DiscreteDenoiser(
        weighting_config={"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
        scaling_config={"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"}, num_idx=1000,
        discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"})
    out = {}
    for name, gcfg in (("cfg3", {"target": "sgm.modules.diffusionmodules.guiders.ScheduledCFGImgTextRef", "params": {"scale": 7.5, "scale_im": 3.5}}),
                       ("cfg2", {"target": "sgm.modules.diffusionmodules.guiders.VanillaCFGImgRef", "params": {"scale": 7.5}})):
        smp = ns.sampling.EulerEDMSampler(discretization_config={"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
                                          num_steps=50, guider_config=gcfg, device="cpu")
        b, n = 1, 2
        x = W.tensor("x", (b, 4, 8, 8), seed=8)
        c = {"crossattn": W.tensor("c_ctx", (b + b * n, 7, 16), seed=8), "vector": W.tensor("c_vec", (b + b * n, 12), seed=8)}
        uc = {"crossattn": W.tensor("uc_ctx", (b + b * n, 7, 16), seed=8), "vector": W.tensor("uc_vec", (b + b * n, 12), seed=8)}
        denoiser = lambda inp, sigma, cc: den(dummy_network, inp, sigma, cc)  # noqa: E731
        res, _ = smp(denoiser, x.clone(), c, uc=uc, num_steps=12)
        out[name] = res
        if name == "cfg3":
            out.update(x=x, **{"c_" + k: v for k, v in c.items()}, **{"uc_" + k: v for k, v in uc.items()})
            out["sigmas50"] = smp.discretization(50, device="cpu")
            out["sigmas12"] = smp.discretization(12, device="cpu")
            out["table"] = den.sigmas
            xin, sin, cin = smp.guider.prepare_inputs(x, torch.full((b,), 3.3), c, uc)
            out["prep_ctx"], out["prep_vec"] = cin["crossattn"], cin["vector"]
            d1, _, _, _ = den(dummy_network, xin, sin, cin)
            out["denoised_first"] = d1
    npz("sampler", **out)


def case_sdxl_keys():
    """state_dict names/shapes of the full SDXL-config UNet (configs/train_co3d_concept.yaml:27-54), built on the meta device."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(refshim.REF_ROOT, "configs", "train_co3d_concept.yaml")))
    params = cfg["model"]["params"]["network_config"]["params"]
    with torch.device("meta"):
        net = ns.openaimodel.UNetModel(**params)
    listing = keyfile("unet_sdxl", net)
    print("sdxl keys:", len(listing), "params(M):", sum(int(np.prod(s)) for s in listing.values()) / 1e6)


def _data_co3d_functions():
    """The camera functions of sgm/data/data_co3d.py (:27-185), compiled from the reference file in place; the module itself
    cannot be imported here (pytorch_lightning / torchvision / pytorch3d.implicitron are absent)."""
    from cd360 import cameras as cams
    src = open(os.path.join(refshim.REF_ROOT, "sgm", "data", "data_co3d.py")).read()
    names = ("intersect_skew_line_groups", "intersect_skew_lines_high_dim", "_point_line_distance", "compute_optical_axis_intersection",
             "normalize_cameras", "centerandalign", "square_bbox")
    wanted = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(wanted) == len(names)
    env = {"torch": torch, "np": np, "join_cameras_as_batch": cams.join_cameras_as_batch, "Rotate": cams.Rotate, "Translate": cams.Translate}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), "data_co3d.py", "exec"), env)
    return env


def _off_centre_rig(n, seed):
    """A ring of cameras around a point that is NOT the origin, at a scale that is NOT 1, with perturbed look-at directions and
    principal points: what normalize_cameras exists to fix."""
    g = torch.Generator().manual_seed(seed)
    base = synth.ring_cameras(n, seed=seed)
    cams_ = []
    for i in range(n):
        c = base[i]
        centre = c.get_camera_center()[0] * 3.7 + torch.tensor([0.4, -1.1, 2.3])
        at = torch.tensor([0.4, -1.1, 2.3]) + 0.05 * torch.randn(3, generator=g)
        pp = 0.03 * torch.randn(2, generator=g)
        cams_.append(synth.look_at_camera(centre.tolist(), at=at.tolist(), focal=float(c.focal_length[0, 0]), pp=pp.tolist()))
    from cd360.cameras import join_cameras_as_batch
    return join_cameras_as_batch(cams_)


def case_cameras():
    env = _data_co3d_functions()
    rig = _off_centre_rig(12, seed=21)
    out = {"rig": pack_cameras([rig])[0]}
    new, p_int, p_line, pp, r = env["normalize_cameras"](rig)
    out.update(norm=pack_cameras([new])[0], p_intersect=p_int, p_line_intersect=p_line, pp=pp, r=r)
    aligned = env["centerandalign"]([new[i] for i in range(len(new))])
    out["aligned"] = pack_cameras([aligned])[0]
    boxes = np.array([[10, 20, 211, 300], [0, 0, 64, 64], [5.5, 7.25, 100.5, 31.0]], dtype=np.float64)
    out["bbox_in"] = boxes
    out["bbox_out"] = np.stack([env["square_bbox"](b, padding=0.1) for b in boxes])
    out["bbox_out_int"] = np.stack([env["square_bbox"](b.astype(np.int64), padding=0.0, astype=int) for b in boxes])
    cam1 = new[3]
    for axis in "xyz":
        fn = getattr(ns.cameraray, f"interpolate_translate_interpolate_{axis}axis")
        lst = fn(cam1, -0.2, 0.21, 0.1)
        out[f"interp_{axis}"] = pack_cameras([refshim_join(lst)])[0]
    out["interp_focal"] = pack_cameras([refshim_join(ns.cameraray.interpolatefocal(cam1, 0.8, 1.25, 0.1))])[0]
    npz("cameras", **out)


from make_golden_params import LOSS_CFG, LossDenoiser as _LossDenoiser, loss_inputs  # noqa: E402


def case_loss():
    rns = refshim.import_reference_loss()
    d = loss_inputs()
    out = {}
    loss_fn = rns.loss.StandardDiffusionLossImgRef(**LOSS_CFG)
    den = _LossDenoiser(d)
    torch.manual_seed(77)
    l2, lfg, lbg, lrgb = loss_fn(None, den, lambda batch: {}, d["x0"], d["x_rgb"], d["xr"], None, d["mask"], None, d["opacity"], {})
    out.update(l2=l2, lfg=lfg, lbg=lbg, lrgb=lrgb, **{"seen_" + k: v for k, v in den.seen.items()})
    # no-mask / l1 variants through get_loss directly
    w = torch.tensor([0.5, 2.0, 1.0]).view(-1, 1, 1, 1)
    mo = W.tensor("loss.mo", (3, 4, 16, 16), seed=31)
    l2n, _, _, _ = loss_fn.get_loss(mo, [], [], d["x0"], d["x_rgb"], w, None, None, d["opacity"], [])
    out["l2_nomask"] = l2n
    l1 = rns.loss.StandardDiffusionLossImgRef(type="l1", **LOSS_CFG).get_loss(mo, [], [], d["x0"], d["x_rgb"], w, None, None, d["opacity"], [])
    out["l1"] = l1[0]
    # sigma samplers on a fixed RNG stream
    torch.manual_seed(5)
    out["cubic"] = rns.sigma_sampling.CubicSampling(**LOSS_CFG["sigma_sampler_config"]["params"])(16)
    out["discrete"] = rns.sigma_sampling.DiscreteSampling(**LOSS_CFG["sigma_sampler_config_ref"]["params"])(16)
    out["edm"] = rns.sigma_sampling.EDMSampling()(16)
    npz("loss", **out)


# ---------------------------------------------------------------- importance sampling (dead upstream)
def case_importance_sampling():
    """Raymarcher.importance_sampling (nerfsd_pytorch3d.py:264-306) up to its third-party call: the reference's own arithmetic turns the
    previous block's weights into (bins, pdf, u) and hands them to pytorch3d._C.sample_pdf, which is not installed.  The call is RECORDED
    (its arguments are the fixture), not emulated: what pytorch3d does with them stays parity-unpinned.  Two cases: the weight maps at
    the ray grid's size, and at a quarter of it (the antialiased resize of :269-286)."""
    import pytorch3d
    S = 8
    rm = ns.nerf.Raymarcher(num_samples=S, far_plane=2.5, stratified=False, training=False, near_plane=0.5)
    got = {}

    def record(bins, weights, outputs, eps):
        got.update(bins=bins.clone(), pdf=weights.clone(), u=outputs.clone(), eps=torch.tensor(eps))

    keep = pytorch3d._C.sample_pdf
    ns.nerf._C.sample_pdf = record
    try:
        out = {}
        for tag, hw_prev, num_rays in (("same", 16, 16), ("resized", 16, 64)):
            g = torch.Generator().manual_seed(5 + hw_prev + num_rays)
            pw = torch.rand(2, hw_prev, S, 1, generator=g) ** 3
            pw[0, 3] = 0.0
            pw[1, 5] = -0.01  # cdf = weights + 0.01 sums to 0: the padding branch of :290-293
            rm.importance_sampling(pw, num_rays, S, "cpu")
            out.update({f"{tag}_prev_weights": pw, f"{tag}_bins": got["bins"], f"{tag}_pdf": got["pdf"], f"{tag}_u": got["u"], f"{tag}_eps": got["eps"]})
        npz("importance_sampling", far=torch.tensor(2.0), near=torch.tensor(0.5), **out)
    finally:
        ns.nerf._C.sample_pdf = keep


def refshim_join(lst):
    from cd360.cameras import join_cameras_as_batch
    return join_cameras_as_batch(lst)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate selected cases only: make_golden.py case_cameras ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    case_cameras()
    case_loss()
    case_nerf(False)
    case_nerf(True)
    case_block(False)
    case_block(True)
    case_block_sdxl()
    case_mask_ref()
    case_st_dual()
    case_customforward()
    case_unet()
    case_unet_grads()
    case_sampler()
    case_sdxl_keys()
    case_importance_sampling()
    assert not os.path.exists(os.path.join(refshim.REF_ROOT, "sgm", "__pycache__")), "bytecode leaked into the reference tree"
