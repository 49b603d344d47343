#!/usr/bin/env python
"""Times the step's GEMM shapes on the library named by CD360_LIB (hipGraph-timed): one line per shape.  Used to A/B probe builds of the K loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tools")]
import torch
from bench_gemm import timeit_graph, rnd
from cd360 import ops
tag = os.path.basename(os.environ.get("CD360_LIB", "default")).replace("libcd360_", "").replace(".so", "")
out = []
for name, M, N, K, kw in (("4096^3", 4096, 4096, 4096, {}), ("L2 ff1", 3072, 10240, 1280, {"geglu": True}), ("L2 ff2", 3072, 1280, 5120, {"res": True}),
                          ("L2 qkv", 3072, 3840, 1280, {}), ("L2 c2c", 3072, 1280, 1280, {"res": True}), ("L1 ff1", 12288, 5120, 640, {"geglu": True}),
                          ("A3 q", 196608, 640, 640, {})):
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
    b32 = rnd(N, seed=3)
    if kw.get("geglu"):
        st, ws = ops.row_stats(a), w.float().sum(1).contiguous()
        fn = lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=True)
    elif kw.get("res"):
        r = rnd(M, N, seed=4).to(torch.bfloat16)
        fn = lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)
    else:
        fn = lambda: ops.gemm(a, w, bias=b32)
    us = min(timeit_graph(fn, n=10 if M * N * K > 3e10 else 20) for _ in range(2))
    out.append(f"{name} {us:7.1f}us {2.0 * M * N * K / us * 1e-6:5.0f}TF")
print(f"{tag:8s} | " + " | ".join(out), flush=True)
