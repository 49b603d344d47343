"""The three routes of bench.py (fused / hooks on the blocks / hooks on a submodule) on the same trajectory: how far apart are they?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))
import torch
import bench
from cd360 import synth
from sgm.modules.attention import BasicTransformerBlock
DEV = "cuda"
latent, refs, steps = 32, 6, 4
torch.set_grad_enabled(False)
net = bench.build_model(latent, refs, 50, DEV)
one = [synth.pose_batch(1, refs, seed=100, n_train=50)[0]]
g = torch.Generator(device=DEV).manual_seed(7)
ctx = torch.randn(3, 77, 2048, generator=g, device=DEV).to(torch.bfloat16)
y = torch.randn(3, 2816, generator=g, device=DEV).to(torch.bfloat16)
x0 = torch.randn(1, 4, latent, latent, generator=g, device=DEV)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())

def run(use_graph):
    smp = bench.Sampler(net, one * 3, ctx, y, 50, use_graph=use_graph)
    xs, traj = x0.clone(), []
    for i in range(steps):
        xs = smp.step(xs, i)
        traj.append(xs.clone())
    return traj

fused = run(False)
seen = []
hs = [m.register_forward_hook(lambda mod, i, o: seen.append(1)) for m in net.modules() if isinstance(m, BasicTransformerBlock)]
hooked = run(False)
hooked_g = run(True)
for h in hs: h.remove()
print("hooks fired", len(seen), "blocks", len(hs))
hs = [m.norm1.register_forward_hook(lambda mod, i, o: None) for m in net.modules() if isinstance(m, BasicTransformerBlock)]
strict = run(False)
for h in hs: h.remove()
print("hooked vs fused", [rel(a, b) for a, b in zip(hooked, fused)])
print("hooked graph vs hooked eager", [rel(a, b) for a, b in zip(hooked_g, hooked)])
print("strict vs fused", [rel(a, b) for a, b in zip(strict, fused)])
