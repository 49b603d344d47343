import sys, os
sys.path.insert(0, 'os.path.dirname(os.path.abspath(__file__))'); sys.path.insert(0, 'os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "custom-diffusion360_amd")')
import torch
from bench_gemm import timeit_graph, rnd
from cd360 import ops
for N, H, W, C in ((3, 32, 32, 1280), (3, 64, 64, 640)):
    x = rnd(N, H * W, C, seed=1).to(torch.bfloat16)
    w = rnd(C, C, 3, 3, seed=2, scale=(9 * C) ** -0.5).to(torch.bfloat16)
    bias = rnd(C, seed=3)
    wp, w9 = ops.pack_upsample_conv_weight(w), ops.pack_conv_weight(w)
    xi = x.reshape(N, H, W, C).permute(0, 3, 1, 2)
    def old():
        up = torch.nn.functional.interpolate(xi, scale_factor=2, mode="nearest")
        return ops.conv_igemm(up.permute(0, 2, 3, 1).reshape(N, 4 * H * W, C), w9, bias, N, 2 * H, 2 * W, 9)
    t_old = timeit_graph(old, n=10)
    t_new = timeit_graph(lambda: ops.conv_up2x(x, wp, bias, N, H, W), n=10)
    print(f"upsample+conv {C}->{C} @{H}->{2*H}: interpolate + 3x3 {t_old:7.1f} us | folded {t_new:7.1f} us", flush=True)
