#!/usr/bin/env python
"""A/B of the halo form of the 3 x 3 convolution (cd360_tuning.conv_halo) on the step's shapes: hipGraph-timed, interleaved, with the
ResBlock epilogue (bias + per-image addend + residual + GroupNorm slab statistics)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tools")]
import torch
from bench_gemm import timeit_graph
from cd360 import _lib, ops
dev, BF = "cuda", torch.bfloat16
shapes = [("L2 1280->1280", 3, 32, 32, 1280, 1280), ("L2 2560->1280", 3, 32, 32, 2560, 1280), ("L2 1920->1280", 3, 32, 32, 1920, 1280), ("L2 640->1280", 3, 32, 32, 640, 1280),
          ("L1 640->640", 3, 64, 64, 640, 640), ("L1 1280->640", 3, 64, 64, 1280, 640), ("L1 1920->640", 3, 64, 64, 1920, 640), ("L1 960->640", 3, 64, 64, 960, 640), ("L1 320->640", 3, 64, 64, 320, 640)]
for tag, N, H, W, cin, cout in shapes:
    x = torch.randn(N, H * W, cin, device=dev).to(BF)
    wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(BF)
    bias = torch.randn(cout, device=dev)
    emb = torch.randn(N, cout, device=dev).to(BF)
    res = torch.randn(N, H * W, cout, device=dev).to(BF)
    fn = lambda: ops.conv_igemm(x, wp, bias, N, H, W, 9, emb, res, want_stats=True)
    r = {}
    for rnd in range(2):
        for mode in (0, -1, 1):
            with _lib.tuning(conv_halo=mode):
                r.setdefault(mode, []).append(timeit_graph(fn, n=20))
    fl = 2.0 * N * H * W * 9 * cin * cout
    print(f"conv {tag:14s}: shifted {min(r[0]):7.1f} us ({fl / min(r[0]) / 1e6:5.0f} TF/s) | default {min(r[-1]):7.1f} | halo {min(r[1]):7.1f} us ({fl / min(r[1]) / 1e6:5.0f} TF/s)  {100 * (min(r[1]) / min(r[0]) - 1):+.1f} %", flush=True)
