"""BASELINE config 4 (main.py fine-tune loop, train_co3d_concept.yaml): one optimisation step of the pose parameters at SDXL width
and depth -- 512^2 images (latent 64^2), batch 4, 4 reference views each, train mode (stratified jitter), random-init weights,
synthetic cameras / latents / text context.  Times forward + loss + backward + AdamW(fp32 master) and prints one JSON line with the
per-kernel breakdown (HIP events on the launch stream, as bench.py).  Not the headline metric: a measurement of the training side.

    python tools/bench_train.py [--steps 5] [--warmup 2] [--batch 4] [--views 4] [--latent 64]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--eval-mode", action="store_true", help="no stratified jitter")
    a = ap.parse_args()
    from cd360 import finetune, ops, sampling, synth
    from make_golden_params import LOSS_CFG, SDXL_NETWORK_CONFIG
    from sgm.util import instantiate_from_config
    dev = "cuda"
    torch.manual_seed(0)
    with torch.device(dev):
        net = instantiate_from_config(SDXL_NETWORK_CONFIG)
    net = net.to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():  # the stock init makes the pose path a no-op
        for _, blk in sampling.pose_blocks(net):
            c = blk.pose_emb_layers.weight.shape[0]
            blk.pose_emb_layers.weight.add_(torch.randn(c, 2 * c, generator=g, device=dev).mul_(0.02).to(torch.bfloat16))
            blk.pose_featurenerf.model.decoder.weight.copy_(torch.randn(4, c, generator=g, device=dev).mul_(0.02))
        for m in net.modules():
            if m.__class__.__name__ == "SpatialTransformer":
                m.proj_out.weight.copy_(torch.randn(m.proj_out.weight.shape, generator=g, device=dev).mul_(0.02))
    net.eval() if a.eval_mode else net.train()
    names = finetune.select_trainable(net, "pose")
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)  # configs/train_co3d_concept.yaml:2,7-8
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    b, n, L = a.batch, a.views, a.latent
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    batch = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=dev), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
                 pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=dev),
                 target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=dev),
                 mask=torch.ones(b, 1, L, L, device=dev), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
    losses = []
    for _ in range(a.warmup):
        losses.append(float(finetune.train_step(net, loss_fn, opt, **batch)[0]))
    torch.cuda.synchronize()
    ops.profile_start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses.append(float(finetune.train_step(net, loss_fn, opt, **batch)[0]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    prof = ops.profile_stop()
    kern = {k: {"ms_per_step": round(v["ms"] / a.steps, 3), "launches_per_step": v["n"] // a.steps,
                **({"tflops": round(v["flops"] / v["ms"] / 1e9, 1)} if v.get("flops") else {}),
                **({"gbs": round(v["bytes"] / v["ms"] / 1e6, 1)} if v.get("bytes") else {})} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    print(json.dumps({"metric": "fine-tune optimisation steps/sec (config 4)", "value": round(1.0 / dt, 4), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 2),
                      "steps": a.steps, "warmup": a.warmup, "dtype": "bf16 (+fp32 master weights)", "data": "synthetic",
                      "config": {"workload": f"SDXL UNet {8 * L}^2, batch {b}, {n} reference views, trainkeys=pose, {'eval' if a.eval_mode else 'train (stratified)'} mode",
                                 "trainable_tensors": len(names), "trainable_params": int(sum(p.numel() for p in opt.params))},
                      "losses": [round(x, 5) for x in losses], "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "hip_kernels": kern}))


if __name__ == "__main__":
    main()
