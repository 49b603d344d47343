import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import weights as W
from cd360 import finetune, _lib
from cd360.cameras import unpack_cameras
from make_golden_params import UNET_TINY
from sgm.modules.diffusionmodules.openaimodel import UNetModel
from test_oracle_cpu import unet_grad_loss
DEV = "cuda"
gold = os.path.join(ROOT, "tests", "golden")
g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(gold, "unet_tiny.npz")).items()}
gg = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(gold, "unet_tiny_grads.npz")).items()}
def run(**tune):
    with _lib.tuning(**tune):
        net = UNetModel(**UNET_TINY).eval()
        W.load_into(net, seed=5)
        net = net.to(DEV, torch.bfloat16)
        finetune.select_trainable(net, "pose")
        out, fgs, alphas, rgbs = net(g["x"].to(DEV), timesteps=g["t"].to(DEV), context=g["ctx"].to(DEV), y=g["y"].to(DEV),
                                     pose=unpack_cameras(g["cams"]), input_ref=g["input_ref"].to(DEV), sigmas_ref=g["sigmas_ref"].to(DEV), mask_ref=None)
        unet_grad_loss(out, fgs, rgbs).backward()
        return dict(net.named_parameters())
for tag, tune in (("default", {}), ("no halo", {"conv_halo": 0}), ("one-pass nerf", {"nerf_kernel": 1}), ("both", {"conv_halo": 0, "nerf_kernel": 1})):
    params = run(**tune)
    for k in gg:
        if k.endswith("nviews.weight"):
            got, want = params[k].grad.float().cpu().reshape(-1), gg[k].reshape(-1)
            C = got.numel() - 198
            mx = want.abs().max()
            e = (got - want).abs()
            print(f"{tag:14s} {k.split('.pose')[0]:44s} C={C} rel all {e.max() / mx:.4f} | vf part {e[:C].max() / mx:.4f} (max |want| {want[:C].abs().max():.3e}) | cam part {e[C + 99:].max() / mx:.4f} (max |want| {want[C + 99:].abs().max():.3e})")
print("---- which backward runs")
from cd360 import ops, grad
orig = ops.nerf_mlp_aggregate_bwd
def spy(*a, **k):
    out = orig(*a, **k)
    dl = out[5]
    print("nerf_mlp_aggregate_bwd: Y", tuple(a[4].shape), "scatter", k.get("scatter", True), "sum over views of dlogit: max", float(dl.sum(1).abs().max()), "max |dlogit|", float(dl.abs().max()))
    return out
ops.nerf_mlp_aggregate_bwd = spy
run()
print("---- worst element")
ops.nerf_mlp_aggregate_bwd = orig
for tag, tune in (("default", {}), ("no halo", {"conv_halo": 0})):
    params = run(**tune)
    k = "input_blocks.3.1.transformer_blocks.0.pose_featurenerf.model.nviews.weight"
    got, want = params[k].grad.float().cpu().reshape(-1), gg[k].reshape(-1)
    C = got.numel() - 198
    e = (got - want).abs()
    top = torch.topk(e, 6).indices.tolist()
    print(tag, "C", C, [(i, round(float(got[i]), 5), round(float(want[i]), 5)) for i in top], "max|want|", float(want.abs().max()))
    print("   cam part got[:6]", got[C + 99:C + 105].tolist(), "want", want[C + 99:C + 105].tolist())
