"""Where does the HOST time of one eagerly launched fine-tuning step go?  cProfile over three steps of tools/bench_train.py's set-up
(SDXL size, bs 4, 4 views, latent 64), sorted by own time and by cumulative time; with CD360_LIBRARY_LINEAR=1 the same for the library route."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)


def main():
    from cd360 import finetune, synth
    from make_golden_params import LOSS_CFG
    from sgm.util import instantiate_from_config
    from test_modules_gpu import _sdxl_net
    DEV = "cuda"
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
    for _ in range(2):
        finetune.train_step(net, loss_fn, opt, as_tensors=True, **batch)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(3):
        finetune.train_step(net, loss_fn, opt, as_tensors=True, **batch)
    host = (time.perf_counter() - t0) / 3 * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    print(f"host time to ISSUE one step: {host:.1f} ms; wall incl. the GPU tail: {wall:.1f} ms", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        finetune.train_step(net, loss_fn, opt, as_tensors=True, **batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(40)
    st.print_callers("method 'to' of")
    st.print_callers("torch.empty")


if __name__ == "__main__":
    main()
