"""fine-tuning objective and the references harvest / delta checkpoint around the pose path.
Golden vectors (tests/golden/loss.npz) come from the reference's own StandardDiffusionLossImgRef and sigma samplers
(tests/golden/make_golden.py::case_loss).  CPU only; the all-gather of the harvest is covered in test_shard_gloo.py."""
import os

import numpy as np
import pytest
import torch

from cd360 import finetune
from make_golden_params import LOSS_CFG, LossDenoiser, loss_inputs
from sgm.util import instantiate_from_config

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TARGET = "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef"


@pytest.fixture(scope="module")
def g():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "loss.npz")).items()}


def test_loss_call_matches_reference_including_rng_order(g):
    d = loss_inputs()
    loss_fn = instantiate_from_config({"target": TARGET, "params": LOSS_CFG})
    den = LossDenoiser(d)
    torch.manual_seed(77)
    l2, lfg, lbg, lrgb = loss_fn(None, den, lambda batch: {}, d["x0"], d["x_rgb"], d["xr"], None, d["mask"], None, d["opacity"], {})
    for k in ("noised", "sigmas", "sigmas_ref", "input_ref"):
        assert torch.equal(den.seen[k], g["seen_" + k]), k  # same sigma / noise draws in the same order
    assert torch.allclose(l2, g["l2"], rtol=1e-5, atol=1e-7) and torch.allclose(lfg, g["lfg"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(lbg, g["lbg"], rtol=1e-5, atol=1e-7) and torch.allclose(lrgb, g["lrgb"], rtol=1e-5, atol=1e-7)
    assert lfg.shape == (3, 2) and lbg.shape == (3, 2) and lrgb.shape == (3, 2)


def test_loss_variants_and_samplers_match_reference(g):
    import weights as W
    d = loss_inputs()
    w = torch.tensor([0.5, 2.0, 1.0]).view(-1, 1, 1, 1)
    mo = W.tensor("loss.mo", (3, 4, 16, 16), seed=31)
    loss_fn = instantiate_from_config({"target": TARGET, "params": LOSS_CFG})
    l2n, lfg, lbg, lrgb = loss_fn.get_loss(mo, [], [], d["x0"], d["x_rgb"], w, None, None, d["opacity"], [])
    assert torch.allclose(l2n, g["l2_nomask"], rtol=1e-6) and lfg == [] and lbg == [] and lrgb == []
    l1 = instantiate_from_config({"target": TARGET, "params": dict(LOSS_CFG, type="l1")}).get_loss(mo, [], [], d["x0"], d["x_rgb"], w, None, None, d["opacity"], [])
    assert len(l1) == 2 and torch.allclose(l1[0], g["l1"], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        instantiate_from_config({"target": TARGET, "params": dict(LOSS_CFG, type="lpips")})
    torch.manual_seed(5)
    assert torch.equal(instantiate_from_config(LOSS_CFG["sigma_sampler_config"])(16), g["cubic"])
    assert torch.equal(instantiate_from_config(LOSS_CFG["sigma_sampler_config_ref"])(16), g["discrete"])
    assert torch.equal(instantiate_from_config({"target": "sgm.modules.diffusionmodules.sigma_sampling.EDMSampling"})(16), g["edm"])


def test_bf16_model_outputs_are_promoted():
    """The HIP UNet returns fp32 eps but bf16 fg/alphas/rgb lists: the loss must come out fp32 and close to the fp32 result."""
    d = loss_inputs()
    loss_fn = instantiate_from_config({"target": TARGET, "params": LOSS_CFG})
    w = torch.ones(3, 1, 1, 1)
    args = lambda cast: (d["x0"] * 0.9, [cast(d["fg0"]), cast(d["fg1"])], [cast(d["rgb0"]), cast(d["rgb1"])], d["x0"], d["x_rgb"], w, d["mask"], None,
                         d["opacity"], [cast(d["alphas0"]), cast(d["alphas1"])])
    a = loss_fn.get_loss(*args(lambda t: t))
    b = loss_fn.get_loss(*args(lambda t: t.bfloat16()))
    for x, y in zip(a, b):
        assert y.dtype == torch.float32 and torch.allclose(x, y, rtol=3e-2, atol=1e-4)


def test_combine_losses_follows_engine_forward():
    l2, lfg, lbg, lrgb = torch.tensor([1.0, 2.0, 3.0]), torch.tensor([[0.1, 0.3], [9.0, 9.0], [0.2, 0.2]]), torch.ones(3, 2) * 0.01, torch.tensor([[0.5, 0.5], [7.0, 7.0], [0.1, 0.3]])
    drop = torch.tensor([1.0, 0.0, 1.0])
    total, parts = finetune.combine_losses(l2, lfg, lbg, lrgb, drop)
    fg, bg, rgb = (0.2 + 0.2) / 2, 0.01, (0.5 + 0.2) / 2
    assert abs(float(total) - (2.0 + 10 * fg + 10 * bg + 5 * rgb)) < 1e-5
    assert abs(parts["loss_fg"] - fg) < 1e-6 and abs(parts["loss_rgb"] - rgb) < 1e-6 and abs(parts["loss"] - 2.0) < 1e-6
    total0, parts0 = finetune.combine_losses(l2, lfg, lbg, lrgb, drop, global_step=0, rgb_predict=False)
    assert abs(float(total0) - 2.0) < 1e-6 and "loss_fg" not in parts0  # step 0 skips the render terms (diffusion.py:229)


def _tiny_unet():
    from make_golden_params import UNET_TINY
    from sgm.modules.diffusionmodules.openaimodel import UNetModel
    return UNetModel(**UNET_TINY)


def test_select_trainable_matches_reference_rules():
    net = _tiny_unet()
    pose = finetune.select_trainable(net, "pose")
    assert pose and all("pose" in n for n in pose)
    assert sum(p.numel() for n, p in net.named_parameters() if p.requires_grad) == sum(p.numel() for n, p in net.named_parameters() if "pose" in n)
    pa = finetune.select_trainable(net, "poseattn")
    blocks = {n.split(".pose")[0] for n in pose}
    for n, p in net.named_parameters():
        want = ("pose" in n) or (any(b in n for b in blocks) and ("attn1" in n or "attn2" in n))
        if "transformer_blocks" in n or "pose" in n:
            assert p.requires_grad == want, n
        else:
            assert not p.requires_grad, n
    assert len(pa) > len(pose)
    assert len(finetune.select_trainable(net, "all")) == len(list(net.parameters()))
    with pytest.raises(ValueError):
        finetune.select_trainable(net, "nope")


def test_delta_checkpoint_round_trip():
    """main.py:611-624 -> sgm/util.py:227-240: pose parameters + references travel, raymarcher buffers and the rest do not."""
    import weights as W
    src, dst = _tiny_unet(), _tiny_unet()
    W.load_into(src, seed=8)
    W.load_into(dst, seed=9)
    names = [n for n, m in src.named_modules() if hasattr(m, "pose_emb_layers")]
    assert names
    for i, n in enumerate(names):
        mod = src.get_submodule(n)
        mod.register_buffer("references", torch.full((3, 4, mod.pose_emb_layers.weight.shape[0]), float(i + 1)))
    full = {"model.diffusion_model." + k: v for k, v in src.state_dict().items()}
    delta = finetune.delta_state_dict(full, embeds=[torch.zeros(1, 8), torch.ones(1, 8)])
    assert all(("pose" in k or "references" in k or k == "embed") for k in delta) and not any("raymarcher" in k for k in delta)
    assert any(k.endswith("pose_emb_layers.weight") for k in delta) and any(k.endswith("plane_coefs.0.weight") for k in delta)
    assert sum(k.endswith(".references") for k in delta) == len(names) and len(delta["embed"]) == 2
    unexpected = finetune.load_delta_state_dict(dst, delta)
    assert unexpected == []
    for n in names:
        a, b = src.get_submodule(n), dst.get_submodule(n)
        assert torch.equal(a.references, b.references) and torch.equal(a.pose_emb_layers.weight, b.pose_emb_layers.weight)
        assert torch.equal(a.pose_featurenerf.model.decoder.weight, b.pose_featurenerf.model.decoder.weight) if hasattr(a, "pose_featurenerf") else True
    # non-pose weights untouched
    k = "input_blocks.0.0.weight"
    assert not torch.equal(src.state_dict()[k], dst.state_dict()[k])
    with pytest.raises(KeyError):
        finetune.load_delta_state_dict(_tiny_unet(), {k: v for k, v in delta.items() if not k.endswith(names[0] + ".references")})


def test_reference_hooks_record_only_pose_free_calls():
    """diffusion.py:28-41: the hook keeps out[0] only when the block ran WITHOUT a pose (out[1] is None)."""
    net = _tiny_unet()
    names = [n for n, m in net.named_modules() if hasattr(m, "pose_emb_layers")]
    acts, handles = finetune.register_reference_hooks(net)
    assert len(handles) == len(names)
    blk = net.get_submodule(names[0])
    C = blk.pose_emb_layers.weight.shape[0]
    x = torch.zeros(2, 4, C)
    for h in list(blk._forward_hooks.values()):
        h(blk, (x,), (x + 1, None, None, None, None))
        h(blk, (x,), (x + 2, torch.ones(2, 4, 1), None, None, None))  # rendered call: not recorded
        h(blk, (x,), (x + 3, None, None, None, None))
    assert len(acts[names[0]]) == 2 and float(acts[names[0]][1].mean()) == 3.0
    with pytest.raises(RuntimeError):
        finetune.harvest_references(net, acts)  # the other blocks recorded nothing
    for n in names[1:]:
        c = net.get_submodule(n).pose_emb_layers.weight.shape[0]
        acts[n].append(torch.zeros(4, 4, c))
    refs = finetune.harvest_references(net, acts)
    assert refs[names[0]].shape == (4, 4, C) and torch.equal(blk.references, refs[names[0]])
    finetune.remove_hooks(handles)
    assert not blk._forward_hooks


def test_train_step_updates_fp32_masters_and_reduces_the_loss():
    """finetune.train_step / MasterAdamW with a stand-in network (CPU): only 'pose' parameters move, the bf16 parameters follow
    fp32 master copies (an lr of 1e-3 on bf16 weights near 1 would round away otherwise), and the total loss goes down."""
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.pose_mix = torch.nn.Parameter(torch.ones(4, dtype=torch.bfloat16))
            self.frozen = torch.nn.Parameter(torch.ones(4, dtype=torch.bfloat16))

        def forward(self, x, timesteps=None, context=None, y=None, pose=None, input_ref=None, sigmas_ref=None, mask_ref=None):
            out = x * (self.pose_mix * self.frozen).float().view(1, 4, 1, 1)
            b = x.shape[0]
            s = self.pose_mix.float().mean()
            return out, [torch.full((b, 16, 1), 0.5) * s], [torch.full((b, 16, 4, 1), 0.5) * s], [torch.full((b, 16, 3), 0.5) * s]

    net = Net()
    assert finetune.select_trainable(net, "pose") == ["pose_mix"]
    opt = finetune.MasterAdamW(net.parameters(), lr=1e-3)
    assert len(opt.params) == 1 and opt.master[0].dtype == torch.float32
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    g = torch.Generator().manual_seed(0)
    b = 2
    x = torch.randn(b, 4, 8, 8, generator=g)
    kw = dict(noised=x, timesteps=None, context=None, y=None, pose=None, input_ref=None, sigmas_ref=None, target=0.5 * x,
              target_rgb=torch.zeros(b, 3, 64, 64), w=torch.ones(b, 1, 1, 1), mask=torch.ones(b, 1, 8, 8), opacity=torch.full((b, 1, 64, 64), 0.5))
    first = float(finetune.train_step(net, loss_fn, opt, **kw)[0])
    for _ in range(30):
        last, logged = finetune.train_step(net, loss_fn, opt, **kw)
    assert float(last) < first and {"loss", "loss_fg", "loss_bg", "loss_rgb"} <= set(logged)
    assert torch.equal(net.frozen.detach(), torch.ones(4, dtype=torch.bfloat16)) and net.frozen.grad is None
    assert (opt.master[0] != 1).all() and torch.equal(net.pose_mix.detach(), opt.master[0].to(torch.bfloat16))
    assert (opt.master[0] - 1).abs().max() < 8e-3 * 30  # 30 steps of 1e-3: individually below bf16's spacing at 1, kept by the masters


def test_optimizer_param_groups_follow_configure_optimizers():
    """diffusion.py:310-361: pose parameters at lr; poseattn adds the pose blocks' attn1 / attn2 weights at multiplier * lr."""
    net = _tiny_unet()
    names = [n for n, _ in net.named_parameters()]
    g = finetune.optimizer_param_groups(net, "pose", lr=1e-4)
    assert len(g) == 1 and g[0]["lr"] == 1e-4 and g[0]["names"] == [n for n in names if "pose" in n]
    g = finetune.optimizer_param_groups(net, "poseattn", lr=1e-4, multiplier=0.05)
    assert len(g) == 2 and abs(g[1]["lr"] - 5e-6) < 1e-12
    pose_blocks = {n.split(".pose")[0] for n in names if "pose" in n}
    assert g[1]["names"] and all(("attn1" in n or "attn2" in n) and "pose" not in n and any(b in n for b in pose_blocks) for n in g[1]["names"])
    assert sorted(g[0]["names"] + g[1]["names"]) == sorted(finetune.select_trainable(net, "poseattn"))
    g = finetune.optimizer_param_groups(net, "all", lr=1e-4)
    assert sorted(g[0]["names"] + g[1]["names"]) == sorted(names)
    finetune.select_trainable(net, "poseattn")
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "poseattn", lr=1e-4), lr=1e-4)
    assert [pg["lr"] for pg in opt.opt.param_groups] == [1e-4, 1e-4 * 0.05]
    assert len(opt.params) == len(finetune.select_trainable(net, "poseattn")) == sum(len(pg["params"]) for pg in opt.opt.param_groups)


def test_combine_losses_as_tensors_is_the_same_total_without_host_reads():
    """as_tensors=True (graph capture / no mid-step synchronisation): logged terms are 0-d tensors, the `loss_rgb.mean() > 0` test of
    diffusion.py:238 becomes arithmetic -- identical totals and gradients, including the all-zero rgb case the reference skips."""
    l2 = torch.tensor([1.0, 2.0, 3.0], requires_grad=True)
    lfg, lbg = torch.tensor([[0.1, 0.3], [9.0, 9.0], [0.2, 0.2]]), torch.ones(3, 2) * 0.01
    drop = torch.tensor([1.0, 0.0, 1.0])
    for e0 in (torch.tensor([[0.7, 0.7], [2.6, 2.6], [0.3, 0.55]]), torch.zeros(3, 2)):  # loss_rgb is a (masked) mean of squared errors (loss.py)
        e = e0.clone().requires_grad_(True)
        t0, p0 = finetune.combine_losses(l2, lfg, lbg, e * e, drop)
        t1, p1 = finetune.combine_losses(l2, lfg, lbg, e * e, drop, as_tensors=True)
        assert torch.equal(t0.detach(), t1.detach()) and all(torch.is_tensor(v) and v.dim() == 0 and not v.requires_grad for v in p1.values())
        assert all(abs(float(p1[k]) - v) < 1e-7 for k, v in p0.items()) and float(p1["loss_rgb"]) == p0.get("loss_rgb", 0.0)
        g0 = torch.autograd.grad(t0, e, allow_unused=True)[0]  # None when the reference's branch skipped the term
        g1 = torch.autograd.grad(t1, e)[0]
        assert torch.equal(g1, torch.zeros_like(g1) if g0 is None else g0)


def test_patch_and_depth_jitter_bounds_are_cached_and_device_side_form_equals_host_form():
    """cd360.nerf.patch_positions / depth_samples with jitter: the host form (jitter drawn on the CPU generator, as the reference) and the
    device form (bounds cached per device, jitter combined where it lives: no host tensor inside a captured step) are the same numbers."""
    from cd360 import nerf
    g = torch.Generator().manual_seed(5)
    for r in (8, 16, 32):
        j = torch.rand(r + 1, generator=g)
        edges = torch.linspace(1, -1, r + 1)
        center = (edges[1:] + edges[:-1]) / 2.0
        upper, lower = torch.cat([center, edges[-1:]], -1), torch.cat([edges[:1], center], -1)
        want = (lower + (upper - lower) * j)[:-1]  # utils_cameraray.py:121-140
        got = nerf.patch_positions(r, "cpu", j)
        assert torch.equal(got, want) and got.shape == (r,)
        assert torch.equal(nerf.patch_positions(r, "cpu"), (edges[:-1] + edges[1:]) / 2)
    jd = torch.rand(64, 25, generator=g)
    t, d = nerf.depth_samples(24, 2.0, 0.0, "cpu", 64, jd)
    t2, d2 = nerf.depth_samples(24, 2.0, 0.0, "cpu", 64, jd)
    assert t.shape == (64, 24) and torch.equal(t, t2) and torch.equal(d, d2) and (d > 0).all() and (t[:, 1:] > t[:, :-1]).all()
    assert ("depth_bounds", 24, 2.0, 0.0, "cpu") in nerf._grid_cache
