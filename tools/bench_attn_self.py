"""Self-attention (attn1, attention.py:609-619) on the whole-tile kernel at the cfg-B shapes, as the transformer blocks call it: q | k | v are
the three column slices of one merged projection output, q pre-scaled.  hipGraph-timed; CD360_LIB=<other build> for a same-box A/B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from bench_gemm import timeit_graph
from cd360 import _lib, ops

dev = "cuda"
print("library:", _lib.LIB_PATH, flush=True)
for rnd in range(2):
    for name, b, H, n in (("L1 64^2", 3, 10, 4096), ("L2 32^2", 3, 20, 1024)):
        inner = H * 64
        qkv = torch.randn(b, n, 3 * inner, device=dev).to(torch.bfloat16)
        run = lambda: ops.attention(qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:], H, n, prescaled=True)
        fl = 4.0 * b * H * n * n * 64
        line, outs = f"attn_self {name} b{b} H{H} N{n}:", {}
        for variant in (-1, 1, 2, 3):  # cd360_tuning.attn_self: by shape / 4 waves x 32 queries / 8 x 64 / 8 x 32 (round 5)
            _lib.set_tuning(attn_self=variant)
            us = timeit_graph(run, n=20)
            outs[variant] = run().clone()
            line += f" | {variant}: {us:7.1f} us {fl / us / 1e6 / 2500:.3f}"
        _lib.set_tuning(attn_self=-1)
        line += " | bit-identical to variant 1: " + str({v: bool(torch.equal(outs[v], outs[1])) for v in (2, 3)})
        print(line, flush=True)
