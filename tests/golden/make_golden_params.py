"""Constructor parameter sets shared by the golden generator and the tests (data only)."""

# reduced-width/depth UNet used for the `unet_tiny` golden vector
UNET_TINY = dict(in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2],
                 num_head_channels=64, use_linear_in_transformer=True, transformer_depth=[1, 1], context_dim=32, adm_in_channels=16,
                 num_classes="sequential", use_checkpoint=False, spatial_transformer_attn_type="softmax-xformers",
                 image_cross_blocks=[0, 1, 2], rgb=True, far=2, num_samples=4, not_add_context_in_triplane=False, rgb_predict=True,
                 add_lora=False, average=False, use_prev_weights_imp_sample=True, stratified=True, imp_sampling_percent=0.9)

# the `network_config` node of the reference's only config (configs/train_co3d_concept.yaml:27-54) lives in the package
from cd360.configs import SDXL_NETWORK_CONFIG  # noqa: E402,F401


# ---- training-loss case (tests/golden/loss.npz) ----
import torch  # noqa: E402

import weights as W  # noqa: E402

LOSS_CFG = {"sigma_sampler_config": {"target": "sgm.modules.diffusionmodules.sigma_sampling.CubicSampling", "params": {
    "num_idx": 1000, "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}}},
    "sigma_sampler_config_ref": {"target": "sgm.modules.diffusionmodules.sigma_sampling.DiscreteSampling", "params": {
        "num_idx": 50, "discretization_config": {"target": "sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization"}}}}


def loss_inputs():
    """Deterministic stand-ins for one training batch (b=3, latent 16x16, image 128x128, pose blocks at r=8 and r=4, S=4)."""
    b, S = 3, 4
    t = lambda name, shape, **k: W.tensor("loss." + name, shape, seed=31, **k)
    d = {"x0": t("x0", (b, 4, 16, 16)), "x_rgb": t("x_rgb", (b, 3, 128, 128)).clamp(-1, 1), "xr": t("xr", (b, 2, 4, 16, 16))}
    d["mask"] = (t("mask", (b, 1, 16, 16)) > -0.3).float()
    d["opacity"] = torch.sigmoid(4 * t("opacity", (b, 1, 128, 128)))
    d["opacity"][:, :, :40] = 0.02
    d["drop_im"] = torch.tensor([1.0, 0.0, 1.0])
    for i, r in enumerate((8, 4)):
        d[f"fg{i}"] = torch.sigmoid(t(f"fg{i}", (b, r * r, 1))) * 1.1 - 0.05
        d[f"alphas{i}"] = torch.sigmoid(t(f"alphas{i}", (b, r * r, S, 1)))
        d[f"rgb{i}"] = torch.sigmoid(t(f"rgb{i}", (b, r * r, 3)))
    return d


class LossDenoiser:
    """Stands in for DiscreteDenoiser in the loss's __call__: returns fixed 'network' outputs and records what it was given."""

    def __init__(self, d):
        self.d, self.seen = d, {}

    def __call__(self, network, noised, sigmas, cond, **kw):
        self.seen = {"noised": noised.clone(), "sigmas": sigmas.clone(), "sigmas_ref": kw["sigmas_ref"].clone(), "input_ref": kw["input_ref"].clone()}
        out = 0.9 * noised / (1 + sigmas.view(-1, 1, 1, 1))
        return out, [self.d["fg0"], self.d["fg1"]], [self.d["alphas0"], self.d["alphas1"]], [self.d["rgb0"], self.d["rgb1"]]

    def w(self, sigma):
        return sigma ** -2.0


# ---- SDXL-width pose block (tests/golden/block_sdxl.npz) ----
SDXL_BLOCK_WIDTHS = ((640, 10), (1280, 20))  # (channels, heads) of the two pose-block levels of the shipped config


def sdxl_block_inputs(C):
    """Inputs of make_golden.py::case_block_sdxl, regenerated on both sides from names and seeds (the fixture stores the packed cameras
    and the reference's outputs only): x [1, 64, C], text context [1, 77, 2048], reference features [2, 64, C], pose (target + 2 views)."""
    from cd360 import synth
    r, n, b, T, cd = 8, 2, 1, 77, 2048
    return (W.tensor(f"sdxl{C}.x", (b, r * r, C), seed=6), W.tensor(f"sdxl{C}.ctx", (b, T, cd), seed=6),
            W.tensor(f"sdxl{C}.cref", (b * n, r * r, C), seed=6), synth.pose_batch(b, n, seed=14 + C // 640))
