"""cd360.prefetch.WeightPrefetcher (`-m gpu`): a step captured with the weight prefetcher armed replays to the same bits as the step captured
without it (the side branch only READS weights), the touch kernels are in the graph, and arming outside a capture / leaving it armed cannot
leak into eager launches."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))

pytestmark = pytest.mark.gpu


def test_prefetched_capture_replays_bit_identical_and_disarms():
    from cd360 import ops
    from cd360.prefetch import WeightPrefetcher
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    a = torch.randn(3072, 1280, generator=g, device=dev).to(torch.bfloat16)
    ws = [(torch.randn(1280, 1280, generator=g, device=dev) * 1280 ** -0.5).to(torch.bfloat16) for _ in range(6)]
    big = (torch.randn(5120, 1280, generator=g, device=dev) * 1280 ** -0.5).to(torch.bfloat16)

    def step(x):
        for w in ws:
            x = ops.gemm(x, w, res=x)
        return ops.gemm(ops.gemm(x, big), big.t().contiguous(), res=x)

    want = step(a)
    outs = []
    for pf in (None, WeightPrefetcher(dev, lag=2, wgs=16)):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(a)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if pf is None:
                out = step(a)
            else:
                with pf:
                    out = step(a)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        outs.append(out.clone())
    assert torch.equal(outs[0], want) and torch.equal(outs[1], want)
    # disarmed: eager launches afterwards enqueue nothing on the side stream and still agree
    assert torch.equal(step(a), want)
