"""Weight prefetcher of a captured step (csrc/prefetch.hip, include/cd360_hip.h): while a step is being captured into a hipGraph, every
GEMM-family launch also enqueues -- on a side stream forked into the same capture -- a small kernel that touches that launch's weights as
soon as the launch `lag` positions earlier has finished.  In the replayed graph the weights arrive in the Infinity Cache beside the
preceding launches instead of being fetched from HBM by the launch that needs them.

    pf = WeightPrefetcher(device)
    with torch.cuda.graph(g):
        with pf:                      # forks the side stream, arms the C side; on exit disarms and joins
            out = step()
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check


class WeightPrefetcher:
    def __init__(self, device, lag: int = 2, wgs: int = 256, min_bytes: int = 1 << 18):
        self.device, self.lag, self.wgs, self.min_bytes = torch.device(device), int(lag), int(wgs), int(min_bytes)
        self.sink = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.side = torch.cuda.Stream(device=self.device)

    def __enter__(self):
        self.side.wait_stream(torch.cuda.current_stream(self.device))  # fork: the side stream joins the capture
        # armed for THIS capture's stream only: launches on other streams (another sampler's capture) are not touched
        self.main = torch.cuda.current_stream(self.device).cuda_stream
        check(_lib.load().cd360_prefetch_arm_on(ctypes.c_void_p(self.main), ctypes.c_void_p(self.side.cuda_stream), self.lag, self.wgs,
                                                self.min_bytes, ctypes.c_void_p(self.sink.data_ptr())), "cd360_prefetch_arm_on")
        return self

    def __exit__(self, *exc):
        check(_lib.load().cd360_prefetch_disarm_on(ctypes.c_void_p(self.main)), "cd360_prefetch_disarm_on")
        torch.cuda.current_stream(self.device).wait_stream(self.side)  # join
        return False
