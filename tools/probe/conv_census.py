"""Every convolution launch of one steady denoise step with its shape and its GPU time (events around each call, eager launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch, bench
from cd360 import synth, ops
from cd360.job import Sampler
dev = torch.device("cuda", 0)
net = bench.build_model(128, 50, 50, dev)
pose = [synth.pose_batch(1, 50, seed=100, n_train=50)[0]] * 3
g = torch.Generator(device=dev).manual_seed(7)
ctx = torch.randn(3, 77, 2048, generator=g, device=dev).to(torch.bfloat16)
y = torch.randn(3, 2816, generator=g, device=dev).to(torch.bfloat16)
x = torch.randn(1, 4, 128, 128, generator=g, device=dev)
smp = Sampler(net, pose, ctx, y, 50, use_graph=False)
x = smp.step(x, 0); x = smp.step(x, 1)
rec = []
def wrap(name, fn, shape_of):
    def inner(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record()
        rec.append((name, shape_of(*a, **k), e0, e1))
        return out
    return inner
ops.conv_igemm = wrap("conv_igemm", ops.conv_igemm, lambda x_, w_, b_, N, H, W, taps=9, emb=None, res=None, want_stats=False, stride=1, **k: (N, H, W, x_.shape[-1], w_.shape[0], taps, stride))
ops.conv_up2x = wrap("conv_up2x", ops.conv_up2x, lambda x_, w_, b_, N, H, W: (N, H, W, x_.shape[-1], w_.shape[1]))
ops.out_conv4 = wrap("out_conv4", ops.out_conv4, lambda x_, w_, b_, N, H, W: (N, H, W, x_.shape[-1], 4))
import sgm.modules.diffusionmodules.util as U
x = smp.step(x, 2)
torch.cuda.synchronize()
from collections import OrderedDict
agg = OrderedDict()
for name, shp, e0, e1 in rec:
    a = agg.setdefault((name, shp), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3
tot = 0
for (name, shp), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print(f"{name:10s} {str(shp):38s} x{n:2d}  {us / n:8.1f} us each  {us:8.1f} us")
print("total", round(tot, 1), "us")
