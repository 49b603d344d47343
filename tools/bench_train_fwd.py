"""Forward pass of the fine-tuning step (BASELINE config 4 shapes: 512^2, batch 4, 4 reference views, dual stream, stratified jitter).
Backward kernels are a later-round row; this measures the forward half: main stream (grad path in the reference) + the no-grad
reference stream over b*n latents + all 12 FeatureNeRF renders EVERY step (no caching in training, attention.py:851-868)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch

from cd360 import ops, synth
from make_golden_params import SDXL_NETWORK_CONFIG
from sgm.util import instantiate_from_config

dev = torch.device("cuda")
b, n, L = int(os.environ.get("B", 4)), int(os.environ.get("NREF", 4)), 64
torch.manual_seed(0)
with torch.device(dev):
    net = instantiate_from_config(SDXL_NETWORK_CONFIG)
net = net.to(torch.bfloat16).train()
g = torch.Generator(device=dev).manual_seed(1)
with torch.no_grad():
    for m in net.modules():
        if hasattr(m, "pose_emb_layers"):
            m.pose_featurenerf.model.decoder.weight.copy_(torch.randn(m.pose_featurenerf.model.decoder.weight.shape, generator=g, device=dev) * 0.02)
pose = synth.pose_batch(b, n, seed=5)
x = torch.randn(b, 4, L, L, generator=g, device=dev)
xr = torch.randn(b, n, 4, L, L, generator=g, device=dev)
ctx = torch.randn(b + b * n, 77, 2048, generator=g, device=dev)
y = torch.randn(b + b * n, 2816, generator=g, device=dev)
t = torch.full((b,), 500.0, device=dev)
sref = torch.full((b,), 100.0, device=dev)


@torch.no_grad()
def step():
    return net(x, timesteps=t, context=ctx, y=y, pose=pose, input_ref=xr, sigmas_ref=sref, mask_ref=None)


for _ in range(2):
    out = step()
torch.cuda.synchronize()
assert torch.isfinite(out[0]).all() and len(out[1]) == 12
ops.profile_start()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
prof = ops.profile_stop()
print(f"train-forward step (b={b}, n={n}, latent {L}^2, dual stream, 12 renders): {dt * 1e3:.1f} ms  ->  {1 / dt:.2f} fwd steps/s")
print({k: round(v['ms'] / K, 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])})
