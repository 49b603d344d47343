"""Host cost of one fine-tuning Linear (forward + backward) through the C++ autograd node and through the Python one: tiny operands, so the
GPU is never the limit; 2000 forward + backward pairs each, interleaved."""
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "custom-diffusion360_amd")]
import torch
from cd360 import ops, routes
x = torch.randn(4, 64, 128, device="cuda").to(torch.bfloat16).requires_grad_(True)
w = torch.randn(128, 128, device="cuda").to(torch.bfloat16)
b = torch.randn(128, device="cuda").to(torch.bfloat16)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        y = ops.linear(x, w, b)
        y.backward(y)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for r in range(3):
    with routes.override(no_host_glue=False): a = run(2000)
    with routes.override(no_host_glue=True): p = run(2000)
    t0 = time.perf_counter()
    for _ in range(2000):
        y = torch.nn.functional.linear(x, w, b); y.backward(y)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 2000 * 1e6
    print(f"Linear fwd+bwd host cost: C++ node {a:.1f} us | Python node {p:.1f} us | torch F.linear {t:.1f} us")
