#!/usr/bin/env python
"""Which torch-issued (at::native / rocclr) kernels does one STEADY denoise step still launch, and from which line of this package?
Runs the bench's model eagerly, records one cached step under torch.profiler with Python stacks, and prints every device kernel that
does not come from libcd360_hip.so next to the innermost frames inside the repository.  (Round 6)"""
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
x = torch.randn(1, 4, latent, latent, generator=g, device=dev)
smp = Sampler(net, pose, ctx, y, 50, use_graph=False)
x = smp.step(x, 0)
x = smp.step(x, 1)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    x = smp.step(x, 2)
    torch.cuda.synchronize()
rows = collections.OrderedDict()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    for k in ev.kernels:
        if any(s in k.name for s in ("gemm_mfma", "attn_self", "gn_", "concat_channels", "conv_igemm", "cfg_euler", "weight_touch", "rowdot", "volrender", "nerf_", "add_layernorm")):
            continue
        frames = [f for f in (ev.stack or []) if "/repo/" in f or "custom-diffusion360_amd" in f]
        key = (ev.name, k.name[:70], tuple(frames[:3]))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += k.duration
print(f"{'n':>3} {'us':>7}  op | kernel | frames")
tot_n = tot_us = 0
for (op, kn, frames), (n, us) in rows.items():
    tot_n += n
    tot_us += us
    print(f"{n:3d} {us:7.1f}  {op} | {kn} | " + " <- ".join(f.replace(ROOT + "/", "") for f in frames))
print(f"total: {tot_n} torch-issued kernels, {tot_us:.1f} us")
