"""Is one fine-tuning step a deterministic function of its inputs?  The set-up of tests/test_modules_gpu.py::
test_config4_train_step_replayed_from_a_hipgraph_follows_the_eager_steps (eval-mode raymarchers: no jitter), R repetitions of K eager steps
from the SAME start in one process; every step's loss, every trainable gradient and every parameter after the update are fingerprinted
(float64 sum and abs-sum) and compared with repetition 0: prints the first (step, tensor, what) that differs."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
DEV = "cuda"


def fp(t):
    t = t.detach().double()
    return (float(t.sum()), float(t.abs().sum()))


def dirty():
    """What runs before the flaky test in tests/test_modules_gpu.py: the SDXL-size train-mode steps of
    test_config4_sdxl_size_train_step_reduces_the_loss (bs 4, 4 views, latent 64), then everything is freed (the allocator keeps the blocks)."""
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    from test_modules_gpu import _sdxl_net
    net, g = _sdxl_net(seed=41)
    net.train()
    finetune.select_trainable(net, "pose")
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    b, n, L = 4, 4, 64
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    batch = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=DEV), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
                 pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=DEV),
                 target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=DEV),
                 mask=torch.ones(b, 1, L, L, device=DEV), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
    print("dirtying steps:", [round(float(finetune.train_step(net, loss_fn, opt, **batch)[0]), 4) for _ in range(4)], flush=True)


def main(R=8, K=5, DIRTY=0):
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    from test_modules_gpu import _sdxl_net
    if DIRTY:
        dirty()
        import gc
        gc.collect()
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
    params = dict(net.named_parameters())
    ref = None
    for rep in range(R):
        net.load_state_dict(start, strict=False)
        opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)
        rec = []
        for step in range(K):
            total, _ = finetune.train_step(net, loss_fn, opt, **batch)
            torch.cuda.synchronize()
            rec.append(("loss", step, "", float(total)))
            for k in names:
                rec.append(("grad", step, k, fp(params[k].grad)))
            for k in names:
                rec.append(("param", step, k, fp(params[k])))
        if ref is None:
            ref = rec
            print("rep 0 losses:", [round(r[3], 6) for r in rec if r[0] == "loss"], flush=True)
            continue
        def far(a, b_):  # beyond the 1e-6 that the fp32 atomics of the view-logit gradient explain
            if a[0] == "loss":
                return abs(a[3] - b_[3]) > 1e-6 * abs(a[3])
            if a[0] == "param":  # any difference at all, outside the tensors the atomics' rounding noise reaches directly
                return a[3] != b_[3] and "nviews" not in a[2]
            return abs(a[3][1] - b_[3][1]) > 1e-5 * max(abs(a[3][1]), 1e-30) and "nviews" not in a[2]
        diffs = [(a, b_) for a, b_ in zip(ref, rec) if far(a, b_)]
        print(f"rep {rep} losses:", [round(r[3], 6) for r in rec if r[0] == "loss"], "| records differing by more than 1e-5:", len(diffs), flush=True)
        for a, b_ in diffs[:8]:
            print("   ", a[0], "step", a[1], a[2], a[3], "vs", b_[3], flush=True)


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:]))
