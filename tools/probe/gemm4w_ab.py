#!/usr/bin/env python
"""The four-wave 256 x 256 arrangement (gemm_cfg = 9: gemm4w_loop.inc) against the eight-wave one (gemm_cfg = 3) and the sixteen-wave GEGLU
default (7): bit-identity of the outputs (same accumulation order), then hipGraph-timed chains of the launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "custom-diffusion360_amd")]
import torch  # noqa: E402
from cd360 import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
time_only = "time" in sys.argv[1:]


def mk(M, N, K, geglu=False, ln=False, bias=True):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5)
    b = torch.randn(N, generator=g, device=dev) if bias else None
    kw = {}
    if ln:
        gamma, beta = 1 + 0.2 * torch.randn(K, generator=g, device=dev), 0.1 * torch.randn(K, generator=g, device=dev)
        wp, wsum, cb = ops.pack_ln_linear(w, b, gamma, beta)
        kw = dict(bias=cb, ln=(ops.row_stats(a), wsum, 1e-5))
        wk = wp
    else:
        kw = dict(bias=b)
        wk = w.to(torch.bfloat16)
    if geglu:
        perm = ops.geglu_row_order(N // 2, dev)
        wk = wk[perm].contiguous()
        kw = {k: ((v[0], v[1][perm].contiguous(), v[2]) if k == "ln" else (None if v is None else v[perm].contiguous())) for k, v in kw.items()}
        kw["geglu"] = True
    return a, wk, kw


def run(cfg, a, w, kw):
    _lib.set_tuning(gemm_cfg=cfg)
    out = ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    _lib.set_tuning(gemm_cfg=-1)
    return out


def timed(cfg, a, w, kw, n=20, reps=5):
    _lib.set_tuning(gemm_cfg=cfg)
    ops.gemm(a, w, **kw)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            ops.gemm(a, w, **kw)
    _lib.set_tuning(gemm_cfg=-1)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


if "groupm" in sys.argv[1:]:  # token tiles per group of the tile order (L2 reuse inside an XCD)
    for (M, N, K, geglu, ln) in ((4096, 4096, 4096, False, False), (8192, 8192, 8192, False, False), (3072, 10240, 1280, True, True)):
        a, w, kw = mk(M, N, K, geglu, ln)
        row = []
        for gm in (1, 2, 3, 4, 6, 8, 12, 16, 32):
            _lib.set_tuning(gemm_group_m=gm)
            row.append(f"{gm}: {timed(9, a, w, kw, n=10):7.1f}")
        _lib.set_tuning(gemm_group_m=-1)
        print(f"{M}x{N}x{K} geglu={geglu}: group_m -> us  " + " | ".join(row), flush=True)
    sys.exit(0)
if "ksweep" in sys.argv[1:]:  # launch time against K at the FF1 shape: intercept = prologue + epilogue of two tile rounds, slope = the loop
    for K in (64, 128, 256, 640, 1280, 2560):
        M, N = 3072, 10240
        a, w, kw = mk(M, N, K, False, False)
        row = [f"cfg {c}: {timed(c, a, w, kw, n=20):7.1f}" for c in (3, 9)]
        ag, wg, kwg = mk(M, N, K, True, True)
        row += [f"geglu+ln cfg {c}: {timed(c, ag, wg, kwg, n=20):7.1f}" for c in (7, 9)]
        wl, bl = w.contiguous(), kw["bias"].to(torch.bfloat16)
        f = lambda: torch.nn.functional.linear(a, wl, bl)
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                f()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        row.append(f"hipBLASLt: {best:7.1f}")
        print(f"3072x10240x{K}: us  " + " | ".join(row), flush=True)
    sys.exit(0)
if "zeros" in sys.argv[1:]:  # power vs stalls: the same launches on all-zero operands (no toggling in the matrix pipe, the same memory traffic)
    for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (3072, 10240, 1280)):
        fl = 2.0 * M * N * K
        for name, fill in (("random", None), ("zeros", 0.0), ("ones", 1.0)):
            a, w, kw = mk(M, N, K, False, False)
            if fill is not None:
                a.fill_(fill); w.fill_(fill)
            row = [f"cfg {c}: {timed(c, a, w, kw, n=10):8.1f} us" for c in (3, 9)]
            wl, bl = w.contiguous(), kw["bias"].to(torch.bfloat16)
            f = lambda: torch.nn.functional.linear(a, wl, bl)
            f(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(10):
                    f()
            gr.replay(); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
            row.append(f"hipBLASLt: {best:8.1f} us")
            print(f"{M}x{N}x{K} {name}: " + " | ".join(row), flush=True)
    sys.exit(0)
if "whatif" in sys.argv[1:]:  # CD360_LIB = the -DCD360_WHATIF build (tools/probe/whatif_build.sh); results are wrong by construction
    for (M, N, K, geglu, ln) in ((4096, 4096, 4096, False, False), (8192, 8192, 8192, False, False), (3072, 10240, 1280, True, True)):
        a, w, kw = mk(M, N, K, geglu, ln)
        if "zero" in sys.argv[1:]:
            a.zero_(); w.zero_()
        fl = 2.0 * M * N * K
        row = []
        for name, bits in (("full", 0), ("no operand traffic", 0x1000), ("no traffic, no fragment reads", 0x2000), ("no barrier", 0x4000), ("no wait for the pieces", 0x8000), ("no X pieces", 0x10000), ("no W pieces", 0x20000), ("pieces out of range", 0x40000), ("pieces with no lane", 0x80000)):
            _lib.set_tuning(whatif=bits)
            us = timed(9, a, w, kw, n=10)
            row.append(f"{name}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s")
        _lib.set_tuning(whatif=0)
        print(f"{M}x{N}x{K} geglu={geglu}: " + " | ".join(row), flush=True)
    sys.exit(0)
if not time_only:
    for (M, N, K, geglu, ln) in ((256, 256, 64, False, False), (256, 256, 128, False, False), (256, 256, 192, False, False), (512, 768, 256, False, False),
                                 (300, 272, 192, False, False), (1024, 512, 1280, True, True), (3072, 10240, 1280, True, True), (3072, 10240, 1280, False, True),
                                 (2000, 1280, 640, False, False)):
        a, w, kw = mk(M, N, K, geglu, ln)
        ref = run(3, a, w, kw)
        outs = [run(9, a, w, kw) for _ in range(3)]
        same = all(torch.equal(o, outs[0]) for o in outs)
        d = (outs[0].float() - ref.float()).abs().max().item()
        print(f"{M}x{N}x{K} geglu={geglu} ln={ln}: cfg 9 == cfg 3: {torch.equal(outs[0], ref)} (max |d| {d:.3e}, max |ref| {ref.float().abs().max().item():.3f}); repeatable: {same}", flush=True)
for (M, N, K, geglu, ln) in ((4096, 4096, 4096, False, False), (8192, 8192, 8192, False, False), (3072, 10240, 1280, True, True), (3072, 10240, 1280, False, True),
                             (3072, 3840, 1280, False, True), (49152, 1280, 1280, False, False), (6144, 10240, 1280, True, True)):
    a, w, kw = mk(M, N, K, geglu, ln)
    fl = 2.0 * M * N * K
    row = []
    for cfg in ((7, 3, 9) if geglu else (3, 9)):
        us = timed(cfg, a, w, kw, n=10 if M * N * K > 1e11 else 20)
        row.append(f"cfg {cfg}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s")
    if not geglu and not ln:
        wl = w.contiguous()
        bl = kw["bias"].to(torch.bfloat16)
        f = lambda: torch.nn.functional.linear(a, wl, bl)
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                f()
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
        row.append(f"hipBLASLt: {best:8.1f} us {fl / best / 1e6:7.1f} TF/s")
    print(f"{M}x{N}x{K} geglu={geglu} ln={ln}: " + " | ".join(row), flush=True)
