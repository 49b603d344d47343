"""Micro-benchmarks of the cd360 kernels at the bench workload's shapes (cfg-B: latent 128^2, b=3, n=50).  GPU only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch

from cd360 import nerf, ops, synth
from cd360.cameras import pack_cameras

dev = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def attn(tag, b, H, nq, nk):
    q = torch.randn(b, nq, H * 64, device=dev).to(BF)
    k = torch.randn(b, (nk + 7) // 8 * 8, H * 64, device=dev).to(BF)
    v = torch.randn(b, nk, H * 64, device=dev).to(BF)
    us = timeit(lambda: ops.attention(q, k, v, H, nk))
    fl = 4.0 * b * H * nq * nk * 64
    by = 2.0 * (2 * b * nq * H * 64 + 2 * b * nk * H * 64)
    print(f"attn {tag:12s} b{b} H{H} Nq{nq} Nk{nk}: {us:9.1f} us  {fl / us / 1e6:8.1f} TF/s  {by / us / 1e3:8.1f} GB/s", flush=True)


def nerf_block(tag, C, r, n=50, b=3, S=24):
    import weights as W
    shapes = {"model.plane_coefs.0.weight": (C, C + 198), "model.plane_coefs.0.bias": (C,), "model.plane_coefs.2.weight": (C, C),
              "model.plane_coefs.2.bias": (C,), "model.nviews.weight": (1, C + 198), "model.nviews.bias": (1,), "model.decoder.weight": (4, C)}
    w = {k[len("model."):]: v.to(dev) for k, v in W.synth_state_dict(shapes, 1).items()}
    fw = nerf.FusedNerfWeights(w["plane_coefs.0.weight"], w["plane_coefs.0.bias"], w["plane_coefs.2.weight"], w["plane_coefs.2.bias"],
                               w["nviews.weight"], w["nviews.bias"], w["decoder.weight"])
    cams = pack_cameras(synth.pose_batch(1, n, seed=1) * b).to(dev)
    hw = r * r
    xref = torch.randn(b, n, hw, C, device=dev).to(BF)
    xs = nerf.patch_positions(r, dev)
    t, dists = nerf.depth_samples(S, 2.0, 0.0, dev, hw)
    Y, lv = nerf.reference_tables(fw, xref)
    zP = torch.randn(b * n, hw, C, device=dev).to(BF)
    cview = nerf.view_constants(fw, cams)
    us = timeit(lambda: ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk), iters=5, warm=1)
    M = b * n * hw * S
    print(f"nerf {tag}: C{C} r{r} n{n} b{b}: {us:9.1f} us  rows {M / 1e6:.1f}M  {M * C / us / 1e3:8.1f} G(row*ch)/s  mfma {2.0 * M * 99 * C / us / 1e6:7.1f} TF/s", flush=True)
    us2 = timeit(lambda: nerf.reference_tables(fw, xref), iters=5, warm=1)
    print(f"     tables (Y GEMM + lv): {us2:9.1f} us", flush=True)


def gn(tag, N, P, C):
    x = torch.randn(N, P, C, device=dev).to(BF)
    g, bta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    us = timeit(lambda: ops.gn_silu(x, g, bta, 32, 1e-5, True))
    print(f"gn_silu {tag}: N{N} P{P} C{C}: {us:8.1f} us  {3.0 * 2 * N * P * C / us / 1e3:8.1f} GB/s", flush=True)


def attn_bwd(tag, b, H, nq, nk, kv_grad=True):
    q = torch.randn(b, nq, H * 64, device=dev).to(BF)
    k = torch.randn(b, nk, H * 64, device=dev).to(BF)
    v = torch.randn(b, nk, H * 64, device=dev).to(BF)
    do = torch.randn(b, nq, H * 64, device=dev).to(BF)
    o, lse = ops.attention(q, k, v, H, want_lse=True)
    us_f = timeit(lambda: ops.attention(q, k, v, H, want_lse=True))
    us = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, H, need_dkv=kv_grad))
    fl = 4.0 * b * H * nq * nk * 64 * (1.5 + (2.0 if kv_grad else 0.0))
    by = 2.0 * b * H * 64 * (4 * nq + (4 if kv_grad else 2) * nk)
    print(f"attn_bwd {tag}: b{b} H{H} Nq{nq} Nk{nk} dkv={kv_grad}: fwd {us_f:8.1f} us | bwd {us:8.1f} us {fl / us / 1e6:7.1f} TF/s {by / us / 1e3:7.1f} GB/s", flush=True)


def ln(tag, rows, C):
    a, b = torch.randn(rows, C, device=dev).to(BF), torch.randn(rows, C, device=dev).to(BF)
    g, bta = torch.ones(C, device=dev, dtype=BF), torch.zeros(C, device=dev, dtype=BF)
    us = timeit(lambda: ops.add_layernorm(a, b, g, bta, 1e-5), iters=50)
    us1 = timeit(lambda: ops.add_layernorm(a, None, g, bta, 1e-5), iters=50)
    print(f"add_layernorm {tag}: rows {rows} C{C}: a+b {us:7.1f} us {4.0 * 2 * rows * C / us / 1e3:7.1f} GB/s | a only {us1:7.1f} us {2.0 * 2 * rows * C / us1 / 1e3:7.1f} GB/s", flush=True)
    p = torch.randn(rows, 8 * C, device=dev).to(BF)
    us2 = timeit(lambda: ops.geglu(p), iters=50)
    print(f"geglu         {tag}: rows {rows} inner {4 * C}: {us2:7.1f} us {3.0 * 2 * rows * 4 * C / us2 / 1e3:7.1f} GB/s", flush=True)


def conv(tag, N, H, W, cin, cout):
    x = torch.randn(N, H * W, cin, device=dev).to(BF)
    wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(BF)
    bias = torch.randn(cout, device=dev)
    us = timeit(lambda: ops.conv_igemm(x, wp, bias, N, H, W, 9))
    fl = 2.0 * N * H * W * 9 * cin * cout
    xi = x.reshape(N, H, W, cin).permute(0, 3, 1, 2)
    w4 = wp.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    us2 = float("nan") if os.environ.get("CD360_NO_MIOPEN") else timeit(lambda: torch.nn.functional.conv2d(xi, w4, bias.to(BF), padding=1))
    print(f"conv {tag}: N{N} {H}x{W} {cin}->{cout}: cd360 {us:8.1f} us {fl / us / 1e6:7.1f} TF/s | MIOpen {us2:8.1f} us {fl / us2 / 1e6:7.1f} TF/s", flush=True)


def gemm(tag, M, K, N):
    x = torch.randn(1, M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    bias = torch.randn(N, device=dev)
    res = torch.randn(1, M, N, device=dev).to(BF)
    us = timeit(lambda: ops.conv_igemm(x, w, bias, M, 1, 1, 1, None, res))
    bb = bias.to(BF)
    us2 = timeit(lambda: torch.nn.functional.linear(x, w, bb) + res)
    us3 = timeit(lambda: torch.nn.functional.linear(x, w, bb))
    fl = 2.0 * M * K * N
    print(f"gemm {tag}: M{M} K{K} N{N}: cd360(+bias+res) {us:8.1f} us {fl / us / 1e6:7.1f} TF/s | hipBLASLt+add {us2:8.1f} us {fl / us2 / 1e6:7.1f} TF/s | hipBLASLt {us3:8.1f} us {fl / us3 / 1e6:7.1f} TF/s", flush=True)


def conv_tilings():
    """The 320-channel convolutions of the 128^2 level (and the 640-channel one behind the Upsample) on the tilings that fit them:
    1 = 256 x 320 (192 tiles at M = 49152), 5 = 192 x 320 / six waves (256 tiles), 6 = 192 x 320 / twelve waves; interleaved, graph-timed,
    outputs and slab statistics compared bit for bit with tiling 1."""
    from cd360 import _lib
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_gemm import timeit_graph
    for tag, N, H, W, cin, cout in (("L0", 3, 128, 128, 320, 320), ("L0 up", 3, 128, 128, 960, 320), ("L0 cat", 3, 128, 128, 640, 320),
                                    ("L0 640", 3, 128, 128, 640, 640)):
        x = torch.randn(N, H * W, cin, device=dev).to(BF)
        wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(BF)
        bias = torch.randn(cout, device=dev)
        fl = 2.0 * N * H * W * 9 * cin * cout
        ts, outs = {}, {}
        for rep in range(3):
            for cfg in (-1, 1, 5, 6):
                _lib.set_tuning(conv_cfg=cfg)
                ts.setdefault(cfg, []).append(timeit_graph(lambda: ops.conv_igemm(x, wp, bias, N, H, W, 9), n=10, reps=3))
                if rep == 0:
                    o = ops.conv_igemm(x, wp, bias, N, H, W, 9, want_stats=True)
                    outs[cfg] = (o[0].clone(), o[1].float().sum(1).clone())
        _lib.set_tuning(conv_cfg=-1)
        same = {c: (torch.equal(outs[c][0], outs[1][0]), float((outs[c][1] - outs[1][1]).abs().max() / outs[1][1].abs().max())) for c in (5, 6)}
        print(f"conv {tag}: N{N} {H}x{W} {cin}->{cout}: " + " | ".join(f"cfg {c}: {min(v):7.1f} us {fl / min(v) / 1e6:6.0f} TF/s" for c, v in ts.items())
              + f" | vs cfg 1 (output bit-identical, stats rel): {same}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "nerf", "gn", "conv"]
    if "conv_tilings" in which:
        conv_tilings()
        sys.exit(0)
    print("env CD360_ATTN_FAST =", os.environ.get("CD360_ATTN_FAST"))
    if "gemm_tn" in which:  # weight-gradient shapes of the config-4 step
        for (M, N, K) in ((98304, 1280, 1280), (393216, 640, 640), (24576, 1280, 1280), (98304, 1280, 112), (98304, 8, 1280), (1024, 1280, 1280), (4096, 1280, 128)):
            a, b = torch.randn(M, N, device=dev).to(torch.bfloat16), torch.randn(M, K, device=dev).to(torch.bfloat16)
            line = f"gemm_tn M={M:6d} N={N:4d} K={K:4d}:"
            us = timeit(lambda: ops.gemm_tn(a, b), iters=20)
            line += f" {us:7.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF/s)"
            print(line, flush=True)
    if "attn" in which:
        attn("L1 self", 3, 10, 4096, 4096)
        attn("L2 self", 3, 20, 1024, 1024)
        attn("L1 cross", 3, 10, 4096, 77)
        attn("L2 cross", 3, 20, 1024, 77)
        attn("L1 pose", 3, 10, 98304, 77)
        attn("L2 pose", 3, 20, 24576, 77)
    if "gn_graph" in which:  # GroupNorm + SiLU timed inside a hipGraph (the eager loop above measures the host): own statistics pass / producer's slab sums
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from bench_gemm import timeit_graph
        for tag, N, P, C in (("L0", 3, 16384, 320), ("L1", 3, 4096, 640), ("L2", 3, 1024, 1280), ("up", 3, 1024, 2560), ("up0", 3, 16384, 960)):
            x = torch.randn(N, P, C, device=dev).to(BF)
            g, bta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            t = x.float().reshape(N, P // 64, 64, C)
            st = torch.stack([t.sum(2), (t * t).sum(2)], -1).contiguous()
            us = timeit_graph(lambda: ops.gn_silu(x, g, bta, 32, 1e-5, True), n=20)
            us2 = timeit_graph(lambda: ops.gn_silu(x, g, bta, 32, 1e-5, True, tile_stats=st), n=20)
            print(f"gn_silu {tag}: N{N} P{P} C{C}: partial+finalize+apply {us:6.1f} us | finalize+apply {us2:6.1f} us ({2.0 * 2 * N * P * C / us2 / 1e3:6.0f} GB/s)", flush=True)
    if "gn" in which:
        gn("L0", 3, 16384, 320); gn("L1", 3, 4096, 640); gn("L2", 3, 1024, 1280); gn("up", 3, 1024, 2560); gn("up0", 3, 16384, 960)
    if "conv" in which:
        conv("L0", 3, 128, 128, 320, 320); conv("L0 up", 3, 128, 128, 960, 320); conv("L0 ups", 3, 128, 128, 640, 640)
        conv("L1", 3, 64, 64, 640, 640); conv("L1 up", 3, 64, 64, 1920, 640); conv("L1 ups", 3, 64, 64, 1280, 1280)
        conv("L2", 3, 32, 32, 1280, 1280); conv("L2 up", 3, 32, 32, 2560, 1280)
    if "attn_par" in which:  # how the self-attention shapes react to more / fewer workgroups (tile quantisation, waves per SIMD)
        attn("L2 self b3", 3, 20, 1024, 1024); attn("L2 self b6", 6, 20, 1024, 1024); attn("L2 self b12", 12, 20, 1024, 1024)
        attn("L1 self 960wg", 3, 10, 4096, 4096); attn("L1 self 768wg", 3, 8, 4096, 4096); attn("L1 self 1536wg", 6, 8, 4096, 4096)
        attn("L1 self 1920wg", 6, 10, 4096, 4096)
    if "attn_bwd" in which:  # config 4 shapes: batch 4, latent 64^2
        attn_bwd("L1 self", 4, 10, 1024, 1024); attn_bwd("L2 self", 4, 20, 256, 256)
        attn_bwd("L1 text", 4, 10, 1024, 77, False); attn_bwd("L2 text", 4, 20, 256, 77, False)
        attn_bwd("L1 pose", 4, 10, 24576, 77, False); attn_bwd("L2 pose", 4, 20, 6144, 77, False)
    if "ln" in which:
        ln("L1", 12288, 640); ln("L2", 3072, 1280); ln("L2 pose tokens", 3 * 24576, 1280)
    if "conv1" in which:
        conv("L1", 3, 64, 64, 640, 640)
    if "attn1" in which:
        attn("L1 self", 3, 10, 4096, 4096)
    if "attn2" in which:
        attn("L2 self", 3, 20, 1024, 1024)
    if "attn_self" in which:
        attn("L1 self", 3, 10, 4096, 4096); attn("L2 self", 3, 20, 1024, 1024)
    if "nerf1" in which:
        nerf_block("L2", 1280, 32)
    if "gemm" in which:
        gemm("L1 qk", 12288, 640, 1280); gemm("L1 out", 12288, 640, 640); gemm("L1 ff1", 12288, 640, 5120); gemm("L1 ff2", 12288, 2560, 640)
        gemm("L2 qk", 3072, 1280, 2560); gemm("L2 out", 3072, 1280, 1280); gemm("L2 ff1", 3072, 1280, 10240); gemm("L2 ff2", 3072, 5120, 1280)
    if "nerf" in which:
        nerf_block("L2", 1280, 32)
        nerf_block("L1", 640, 64)
