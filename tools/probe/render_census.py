#!/usr/bin/env python
"""Per-kernel census of the RENDER step (the first step of a sampler: FeatureNeRF renders of the twelve pose blocks + one denoise step),
eager launches under torch.profiler: every device kernel, grouped by name, with launch counts and GPU time; torch-issued kernels carry the
innermost frames inside the repository."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from cd360 import synth  # noqa: E402
from cd360.job import Sampler  # noqa: E402

dev = torch.device("cuda", 0)
latent, refs = 128, 50
net = bench.build_model(latent, refs, 50, dev)
pose = [synth.pose_batch(1, refs, seed=100, n_train=50)[0]] * 3
g = torch.Generator(device=dev).manual_seed(7)
ctx = torch.randn(3, 77, 2048, generator=g, device=dev).to(torch.bfloat16)
y = torch.randn(3, 2816, generator=g, device=dev).to(torch.bfloat16)
x0 = torch.randn(1, 4, latent, latent, generator=g, device=dev)
from cd360 import ops  # noqa: E402
smp = Sampler(net, pose, ctx, y, 50, use_graph=False)
for _ in range(2):
    x = smp.step(x0.clone(), 0)
x = smp.step(x, 1)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

for idx in (0, 1):
    ops.profile_start(shapes=True)
    x = smp.step(x0.clone(), idx)
    pr = ops.profile_stop()
    print(f"step {idx}: HIP ops {sum(v['n'] for v in pr.values())} launches, {sum(v['ms'] for v in pr.values()):.2f} ms (event-timed, eager)")
    if idx == 0:
        pr0 = pr
    else:
        for k in sorted(pr0, key=lambda k: -(pr0[k]["ms"] - pr.get(k, {"ms": 0})["ms"])):
            d_ms, d_n = pr0[k]["ms"] - pr.get(k, {"ms": 0})["ms"], pr0[k]["n"] - pr.get(k, {"n": 0})["n"]
            if abs(d_ms) > 0.02 or d_n:
                print(f"   render - steady: {d_ms:8.3f} ms {d_n:5d} launches  {k}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    x = smp.step(x0.clone(), 0)
    torch.cuda.synchronize()
rows = collections.OrderedDict()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    for k in ev.kernels:
        ours = not ("at::native" in k.name or "rocclr" in k.name or "rocblas" in k.name.lower() or "Cijk" in k.name)
        frames = () if ours else tuple(f for f in (ev.stack or []) if "/repo/" in f or "custom-diffusion360_amd" in f)[:2]
        key = (k.name[:90], frames)
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += k.duration
tot_n = sum(r[0] for r in rows.values())
tot_us = sum(r[1] for r in rows.values())
print(f"render step: {tot_n} kernels, {tot_us / 1e3:.2f} ms of kernel time")
for (kn, frames), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{us:9.1f} us {n:5d}  {kn}" + ("  <- " + " <- ".join(f.replace(ROOT + "/", "") for f in frames) if frames else ""))
