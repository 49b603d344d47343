#!/usr/bin/env python
"""FF1 + GEGLU (3072 x 10240 x 1280) with COLD weights: a chain that walks 24 different weight buffers (630 MB: no reuse out of the
256 MB Infinity Cache), per tiling, against the same chain on one buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch
from cd360 import _lib, ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
M, N, K = 3072, 10240, 1280
a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
gamma, beta = 1 + 0.2 * torch.randn(K, generator=g, device=dev), 0.1 * torch.randn(K, generator=g, device=dev)
st = ops.row_stats(a)
perm = ops.geglu_row_order(N // 2, dev)
ws = []
for i in range(24):
    w = torch.randn(N, K, generator=g, device=dev) * K ** -0.5
    b = torch.randn(N, generator=g, device=dev)
    wp, wsum, cb = ops.pack_ln_linear(w, b, gamma, beta)
    ws.append((wp[perm].contiguous(), wsum[perm].contiguous(), cb[perm].contiguous()))
def chain(cfg, asm4, cold, n=24, reps=5):
    _lib.set_tuning(gemm_cfg=cfg, gemm_asm4=asm4)
    def f(i):
        wp, wsum, cb = ws[i if cold else 0]
        return ops.gemm(a, wp, bias=cb, ln=(st, wsum, 1e-5), geglu=True)
    f(0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n):
            f(i)
    _lib.set_tuning(gemm_cfg=-1, gemm_asm4=-1)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best
for rep in range(2):
    for name, cfg, a4 in (("sixteen waves (default)", 7, 0), ("eight waves", 3, 0), ("four waves, generated loop", 9, 1)):
        print(f"{name:28s}: warm {chain(cfg, a4, False):6.1f} us | cold {chain(cfg, a4, True):6.1f} us", flush=True)
