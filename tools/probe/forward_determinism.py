"""Is the fine-tuning FORWARD (and backward) a deterministic function of parameters and inputs?  The set-up of the config-4 graph test,
K optimiser steps to leave the start point, then R times forward + loss + backward on the SAME parameters: distinct values of the loss and
of every gradient fingerprint are counted."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
DEV = "cuda"


def main(K=3, R=60, DIRTY=1):
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    from test_modules_gpu import _sdxl_net
    if DIRTY:
        from train_determinism import dirty
        dirty()
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
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    print("steps:", [round(float(finetune.train_step(net, loss_fn, opt, **batch)[0]), 6) for _ in range(K)], flush=True)
    params = dict(net.named_parameters())
    losses = collections.Counter()
    outs = collections.Counter()
    grads = {k: collections.Counter() for k in names}
    for r in range(R):
        opt.zero_grad()
        out, fgs, alphas, rgbs = net(batch["noised"], timesteps=batch["timesteps"], context=batch["context"], y=batch["y"], pose=batch["pose"],
                                     input_ref=batch["input_ref"], sigmas_ref=batch["sigmas_ref"], mask_ref=None)
        l2, lfg, lbg, lrgb = loss_fn.get_loss(out, fgs, rgbs, batch["target"], batch["target_rgb"], batch["w"], batch["mask"], None, batch["opacity"], alphas)
        total, _ = finetune.combine_losses(l2, lfg, lbg, lrgb, torch.ones(b, device=DEV), as_tensors=True)
        total.backward()
        torch.cuda.synchronize()
        losses[float(total)] += 1
        outs[(float(out.double().sum()), float(out.double().abs().sum()), tuple(float(f.double().sum()) for f in fgs))] += 1
        for k in names:
            gk = params[k].grad.double()
            grads[k][(float(gk.sum()), float(gk.abs().sum()))] += 1
    print(f"{R} forward + backward passes on fixed parameters: distinct losses {dict(losses)}", flush=True)
    print(f"   distinct (eps, fg) fingerprints: {len(outs)}: counts {sorted(outs.values(), reverse=True)}", flush=True)
    multi = {k: sorted(c.values(), reverse=True) for k, c in grads.items() if len(c) > 1}
    print(f"   gradients with more than one fingerprint: {len(multi)} of {len(names)}", flush=True)
    for k, v in list(multi.items())[:40]:
        print("     ", k, v, flush=True)


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:]))
