"""Are the transformer GEMMs of a denoise step slower inside the step than in a warm micro-benchmark because their WEIGHTS are cold?  One
step streams 5.3 GB of weights through a 256 MB Infinity Cache, so every launch finds its weights in HBM, while a loop over one weight
tensor finds them cached.  Times each shape warm (one weight tensor) and cold (a pool of distinct weight tensors larger than the cache,
walked round-robin), the activations the same (warm) tensor in both: 20 launches per hipGraph, best of 5 replays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from bench_gemm import rnd
from cd360 import ops

dev = "cuda"


def graph_time(fns, reps=5, prefetch=None):
    import contextlib
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for f in fns[:3]:
            f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        with (prefetch if prefetch is not None else contextlib.nullcontext()):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(fns))
    return best * 1e3


for name, M, N, K, kw in (("L2 ff1+geglu", 3072, 10240, 1280, "geglu"), ("L2 ff2", 3072, 1280, 5120, "res"), ("L2 out", 3072, 1280, 1280, "res"),
                          ("L2 qkv", 3072, 3840, 1280, "ln"), ("L1 ff1+geglu", 12288, 5120, 640, "geglu")):
    a = rnd(M, K, seed=1).to(torch.bfloat16)
    pool_n = max(24, int(600e6 // (N * K * 2)) + 1)  # > 2x the 256 MB cache
    pool = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(pool_n)]
    b32 = rnd(N, seed=3)
    r = rnd(M, N, seed=4).to(torch.bfloat16)
    st = ops.row_stats(a)
    ws = pool[0].float().sum(1).contiguous()

    def call(w):
        if kw == "geglu":
            return lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=True)
        if kw == "ln":
            return lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5))
        return lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)

    warm = graph_time([call(pool[0])] * 20)
    seq = [call(pool[i % pool_n]) for i in range(max(20, pool_n))]
    cold = graph_time(seq)
    line = f"{name:14s} M={M} N={N} K={K}: warm weights {warm:6.1f} us | cold weights ({pool_n} x {N * K * 2 / 1e6:.1f} MB pool) {cold:6.1f} us | +{(cold / warm - 1) * 100:.0f} %"
    from cd360.prefetch import WeightPrefetcher
    for lag, wgs in ((2, 32), (3, 32), (2, 128)):
        t = graph_time(seq, prefetch=WeightPrefetcher(dev, lag=lag, wgs=wgs))
        line += f" | prefetched lag {lag} x {wgs} WGs {t:6.1f}"
    print(line, flush=True)
    del pool
    torch.cuda.empty_cache()
