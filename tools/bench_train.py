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


def measure(steps=5, warmup=2, batch=4, views=4, latent=64, eval_mode=False, profile=True, library=False, graph=False):
    """One fine-tune configuration: build the SDXL UNet, run `warmup` + `steps` optimisation steps, return the result dict.
    graph=True: the step captured into a hipGraph (finetune.GraphedTrainStep) and replayed -- no per-kernel breakdown then.
    library=True: the round-1 path (cd360.routes.library_linear: every Linear on torch / hipBLASLt) for an A/B on the same box."""
    from cd360 import finetune, ops, sampling, synth
    from make_golden_params import LOSS_CFG, SDXL_NETWORK_CONFIG
    from sgm.util import instantiate_from_config
    from cd360 import routes
    with routes.override(library_linear=bool(library)):
        return _measure(steps, warmup, batch, views, latent, eval_mode, profile and not graph, finetune, ops, sampling, synth, LOSS_CFG,
                        SDXL_NETWORK_CONFIG, instantiate_from_config, graph)


def _measure(steps, warmup, batch, views, latent, eval_mode, profile, finetune, ops, sampling, synth, LOSS_CFG, SDXL_NETWORK_CONFIG, instantiate_from_config,
             graph=False):
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
    net.eval() if eval_mode else net.train()
    names = finetune.select_trainable(net, "pose")
    opt = finetune.MasterAdamW(finetune.optimizer_param_groups(net, "pose", lr=1e-4), lr=1e-4)  # configs/train_co3d_concept.yaml:2,7-8
    loss_fn = instantiate_from_config({"target": "sgm.modules.diffusionmodules.loss.StandardDiffusionLossImgRef", "params": LOSS_CFG})
    b, n, L = batch, views, latent
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    bt = dict(noised=rn(b, 4, L, L), timesteps=torch.full((b,), 500.0, device=dev), context=rn(b + b * n, 77, 2048), y=rn(b + b * n, 2816),
              pose=synth.pose_batch(b, n, seed=3), input_ref=rn(b, n, 4, L, L), sigmas_ref=torch.full((b,), 3.0, device=dev),
              target=rn(b, 4, L, L), target_rgb=rn(b, 3, 8 * L, 8 * L).clamp(-1, 1), w=torch.full((b, 1, 1, 1), 0.7, device=dev),
              mask=torch.ones(b, 1, L, L, device=dev), opacity=torch.sigmoid(3 * rn(b, 1, 8 * L, 8 * L)))
    losses = []
    step = lambda: finetune.train_step(net, loss_fn, opt, **bt)
    if graph:
        step = finetune.GraphedTrainStep(net, loss_fn, opt, bt, warmup=max(warmup, 1), weight_prefetch=bool(os.environ.get("CD360_TRAIN_PREFETCH")))
    for _ in range(warmup):
        losses.append(float(step()[0]))
    torch.cuda.synchronize()
    if profile:
        ops.profile_start(shapes=bool(os.environ.get("CD360_PROFILE_SHAPES")))  # per-shape GEMM rows for the shape table of DESIGN.md
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(float(step()[0]))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = ops.profile_stop() if profile else {}
    kern = {k: {"ms_per_step": round(v["ms"] / steps, 3), "launches_per_step": v["n"] // steps,
                **({"tflops": round(v["flops"] / v["ms"] / 1e9, 1)} if v.get("flops") else {}),
                **({"gbs": round(v["bytes"] / v["ms"] / 1e6, 1)} if v.get("bytes") else {})} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    out = {"metric": "fine-tune optimisation steps/sec (config 4)", "value": round(1.0 / dt, 4), "unit": "steps/s", "ms_per_step": round(dt * 1e3, 2),
           "steps": steps, "warmup": warmup, "launch": "hipGraph replay" if graph else "eager", "dtype": "bf16 (+fp32 master weights)", "data": "synthetic",
           "config": {"workload": f"SDXL UNet {8 * L}^2, batch {b}, {n} reference views, trainkeys=pose, {'eval' if eval_mode else 'train (stratified)'} mode",
                      "trainable_tensors": len(names), "trainable_params": int(sum(p.numel() for p in opt.params))},
           "losses": [round(x, 5) for x in losses], "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "hip_kernels": kern}
    del net, opt, loss_fn, bt, step
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--eval-mode", action="store_true", help="no stratified jitter")
    ap.add_argument("--library", action="store_true", help="every Linear on torch / hipBLASLt (CD360_LIBRARY_LINEAR=1): the A/B partner")
    ap.add_argument("--graph", action="store_true", help="capture the step into a hipGraph and replay it (finetune.GraphedTrainStep)")
    a = ap.parse_args()
    print(json.dumps(measure(a.steps, a.warmup, a.batch, a.views, a.latent, a.eval_mode, True, a.library, a.graph)))


if __name__ == "__main__":
    main()
