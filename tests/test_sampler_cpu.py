"""cd360.sampler (EulerEDMSampler / guiders / DiscreteDenoiser / LegacyDDPMDiscretization) against golden vectors produced by the
reference's own classes around a deterministic dummy network (tests/golden/make_golden.py::case_sampler).  CPU only."""
import os

import numpy as np
import torch

import weights as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def dummy_network(x_in, c_noise, cond, **kw):
    b = x_in.shape[0]
    ctx = cond["crossattn"][:b].float().mean((1, 2)).view(-1, 1, 1, 1)
    vec = cond["vector"][:b].float().mean(1).view(-1, 1, 1, 1)
    pred = 0.3 * torch.tanh(x_in) + 0.001 * c_noise.float().view(-1, 1, 1, 1) / 10 + 0.1 * ctx + 0.05 * vec
    return pred, [], [], [torch.zeros(b, 4, 3)]


def load():
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLD, "sampler.npz")).items()}


def build(guider):
    """through the reference's dotted paths + instantiate_from_config, as sample.py / the YAML do"""
    from sgm.util import instantiate_from_config
    den = instantiate_from_config({"target": "sgm.modules.diffusionmodules.denoiser.DiscreteDenoiser", "params": {
        "num_idx": 1000, "weighting_config": {"target": "sgm.modules.diffusionmodules.denoiser_weighting.EpsWeighting"},
        "scaling_config": {"target": "sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling"},
        "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}}})
    smp = instantiate_from_config({"target": "sgm.modules.diffusionmodules.sampling.EulerEDMSampler", "params": {
        "num_steps": 50, "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"},
        "guider_config": guider, "device": "cpu"}})
    return den, smp


def test_schedule_and_table_match_reference_bitwise():
    g = load()
    den, smp = build({"target": "sgm.modules.diffusionmodules.guiders.ScheduledCFGImgTextRef", "params": {"scale": 7.5, "scale_im": 3.5}})
    assert torch.equal(den.sigmas, g["table"])
    assert torch.equal(smp.discretization(50), g["sigmas50"]) and torch.equal(smp.discretization(12), g["sigmas12"])
    idx = den.sigma_to_idx(torch.tensor([0.03, 3.3, 20.0]))
    assert idx.tolist() == [0, int((g["table"] - 3.3).abs().argmin()), 999]


def test_guider_prepare_and_denoiser_match_reference():
    g = load()
    den, smp = build({"target": "sgm.modules.diffusionmodules.guiders.ScheduledCFGImgTextRef", "params": {"scale": 7.5, "scale_im": 3.5}})
    c = {"crossattn": g["c_crossattn"], "vector": g["c_vector"]}
    uc = {"crossattn": g["uc_crossattn"], "vector": g["uc_vector"]}
    xin, sin, cin = smp.guider.prepare_inputs(g["x"], torch.full((1,), 3.3), c, uc)
    assert xin.shape[0] == 3 and torch.equal(cin["crossattn"], g["prep_ctx"]) and torch.equal(cin["vector"], g["prep_vec"])
    d1 = den(dummy_network, xin, sin, cin)[0]
    assert torch.allclose(d1, g["denoised_first"], atol=1e-6)


def test_full_trajectories_match_reference():
    g = load()
    c = {"crossattn": g["c_crossattn"], "vector": g["c_vector"]}
    uc = {"crossattn": g["uc_crossattn"], "vector": g["uc_vector"]}
    for name, guider in (("cfg3", {"target": "sgm.modules.diffusionmodules.guiders.ScheduledCFGImgTextRef", "params": {"scale": 7.5, "scale_im": 3.5}}),
                         ("cfg2", {"target": "sgm.modules.diffusionmodules.guiders.VanillaCFGImgRef", "params": {"scale": 7.5}})):
        den, smp = build(guider)
        res, rgb = smp(lambda inp, s, cc: den(dummy_network, inp, s, cc), g["x"].clone(), c, uc=uc, num_steps=12)
        assert torch.allclose(res, g[name], atol=2e-5, rtol=1e-5), (name, (res - g[name]).abs().max())
        assert rgb is not None


def test_fused_tail_formula_equals_the_unfused_chain():
    """cfg_euler_update's algebra (what cd360_cfg_euler_step_f32 computes) == denoiser c_out + guider + to_d + Euler, on CPU."""
    from cd360.sampler import ScheduledCFGImgTextRef, cfg_euler_update
    x, eps = W.tensor("x", (2, 4, 8, 8), seed=3), W.tensor("eps", (6, 4, 8, 8), seed=3)
    s, sn = torch.tensor(3.3), torch.tensor(2.9)
    den = torch.cat([x] * 3) - s * eps
    d0 = ScheduledCFGImgTextRef(7.5, 3.5)(den, None)
    want = x + (x - d0) / s * (sn - s)
    assert torch.allclose(cfg_euler_update(x, eps, s, sn, 7.5, 3.5, fused=False), want, atol=1e-6)


def run_product_steps(g, dev, fused):
    """The 12-step cfg3 trajectory of sampler.npz through the product's step function (cd360.sampler.fused_cfg3_euler_step: what
    cd360/job.py's Sampler launches per denoise step), around the golden's dummy network."""
    from cd360 import sampler as S
    den = S.DiscreteDenoiser().to(dev)
    guider = S.ScheduledCFGImgTextRef(7.5, 3.5)
    c = {"crossattn": g["c_crossattn"].to(dev), "vector": g["c_vector"].to(dev)}
    uc = {"crossattn": g["uc_crossattn"].to(dev), "vector": g["uc_vector"].to(dev)}
    x = g["x"].to(dev)
    _, _, cond3 = guider.prepare_inputs(x, x.new_ones(x.shape[0]), c, uc)  # once per image: constant over the trajectory
    sigmas = S.LegacyDDPMDiscretization()(12, device=dev)
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)  # EulerEDMSampler.prepare_sampling_loop (sampling.py:52-66)
    network = lambda x_in, c_noise: dummy_network(x_in, c_noise, cond3)[0]  # noqa: E731
    for i in range(12):
        x = S.fused_cfg3_euler_step(den, network, x, sigmas[i], sigmas[i + 1], 7.5, 3.5, fused=fused)
    return x


def test_product_step_function_walks_the_reference_trajectory():
    """f2 on CPU: the un-fused form of the product's step (same function, fused=False) reproduces the reference's 12-step trajectory;
    tests/test_f_rows_gpu.py runs the SAME function with the HIP kernel on the GPU against the same golden."""
    g = load()
    res = run_product_steps(g, "cpu", fused=False)
    assert torch.allclose(res, g["cfg3"], atol=2e-5, rtol=1e-5), (res - g["cfg3"]).abs().max()
