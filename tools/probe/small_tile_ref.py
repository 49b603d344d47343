#!/usr/bin/env python
"""The narrow GEMMs of the step (N = 1280: C -> C projections, FF2) against the library on the same shapes, hipGraph-timed chains: what a
128 x 128 tiling leaves on the table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch
from cd360 import _lib, ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
def chain(f, n=20, reps=5):
    f(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            f()
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
for (M, N, K) in ((3072, 1280, 1280), (3072, 1280, 2560), (3072, 1280, 5120), (3072, 1280, 10240), (12288, 640, 640), (12288, 640, 2560), (3072, 3840, 1280)):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev)
    r = torch.randn(M, N, generator=g, device=dev).to(torch.bfloat16)
    bl = b.to(torch.bfloat16)
    row = []
    for name, ks in (("mode 0", 0), ("ksplit", 1)):
        _lib.set_tuning(gemm_ksplit=ks)
        row.append(f"cd360 {name}: {chain(lambda: ops.gemm(a, w, bias=b, res=r)):6.1f}")
    _lib.set_tuning(gemm_ksplit=-1)
    row.append(f"cd360 default: {chain(lambda: ops.gemm(a, w, bias=b, res=r)):6.1f}")
    row.append(f"hipBLASLt (bias only): {chain(lambda: torch.nn.functional.linear(a, w, bl)):6.1f}")
    row.append(f"hipBLASLt + residual add: {chain(lambda: torch.nn.functional.linear(a, w, bl) + r):6.1f}")
    print(f"{M}x{N}x{K} us: " + " | ".join(row), flush=True)
