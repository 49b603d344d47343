"""cd360_gemm_bf16 on the GPU: parity against an fp32 torch reference (every epilogue, ragged shapes, repeat-run race screen) and
timing against the library GEMM (F.linear -> hipBLASLt) on the transformer shapes of the SDXL UNet at cfg-B.
    python tools/bench_gemm.py [check] [time] [CFG=1|2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))
import torch
import torch.nn.functional as F

from cd360 import ops
from cd360._lib import ENV  # CD360_* tiling switches -> cd360_set_tuning (the C side reads no environment)

dev = torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def relerr(got, want):
    return ((got.float() - want).abs().max() / want.abs().max().clamp_min(1e-6)).item()


def check_case(M, N, K, bias=False, res=False, ln=False, geglu=False, stats=False, cfg=None, lda=None):
    if cfg:
        ENV["CD360_GEMM_CFG"] = str(cfg)
    else:
        ENV.pop("CD360_GEMM_CFG", None)
    a_full = rnd(M, lda or K, seed=1).to(torch.bfloat16)
    a = a_full[:, :K]
    if ln:  # rows with a mean and a scale, like a residual stream
        a = (a.float() * (0.5 + rnd(M, 1, seed=7).abs()) + 0.7 * rnd(M, 1, seed=8)).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=K ** -0.5)
    b = rnd(N, seed=3) if bias else None
    r = rnd(M, N, seed=4).to(torch.bfloat16) if res else None
    a32 = a.float()
    if ln:
        gamma, beta = 1 + 0.2 * rnd(K, seed=5), 0.1 * rnd(K, seed=6)
        wp, wsum, cb = ops.pack_ln_linear(w, b, gamma, beta)
        want = F.linear(F.layer_norm(a32, (K,), gamma, beta, 1e-5), w, b)
        st = ops.row_stats(a)
        kw = dict(bias=cb, ln=(st, wsum, 1e-5))
        wk = wp
    else:
        want = F.linear(a32, w.to(torch.bfloat16).float(), b)
        kw = dict(bias=b)
        wk = w.to(torch.bfloat16)
    if geglu:
        inner = N // 2
        perm = ops.geglu_row_order(inner, dev)
        wk = wk[perm].contiguous()
        kw = {k: ((v[0], v[1][perm].contiguous(), v[2]) if k == "ln" else (None if v is None else v[perm].contiguous())) for k, v in kw.items()}
        want = want[:, :inner] * F.gelu(want[:, inner:])
    if res:
        want = want + r.float()
    outs = []
    for _ in range(4):  # race screen: identical launches must agree bit for bit
        got = ops.gemm(a, wk, res=r, want_stats=stats, geglu=geglu, **kw)
        outs.append(got)
    torch.cuda.synchronize()
    st_out = None
    if stats:
        st_out = [o[1] for o in outs]
        outs = [o[0] for o in outs]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    e = relerr(outs[0], want)
    msg = f"M={M} N={N} K={K} cfg={cfg or 'auto'} bias={int(bias)} res={int(res)} ln={int(ln)} geglu={int(geglu)} stats={int(stats)}: err {e:.2e} repeat-equal {same}"
    ok = same and e < (1.2e-2 if ln else 8e-3) and bool(torch.isfinite(outs[0]).all())
    if stats:
        o32 = outs[0].float()
        tn = ops.gemm_tile_n(M, N)
        parts = st_out[0].shape[1]
        ws = torch.stack([torch.stack([o32[:, i * tn:(i + 1) * tn].sum(1), (o32[:, i * tn:(i + 1) * tn] ** 2).sum(1)], -1) for i in range(parts)], 1)
        es = relerr(st_out[0], ws)
        msg += f" stats err {es:.2e}"
        ok = ok and es < 1e-4 and all(torch.equal(st_out[0], s) for s in st_out[1:])
    print(("ok   " if ok else "FAIL ") + msg, flush=True)
    return ok


def check_qattn(b, nq, C, K, nk, ln=True, qcfg=None):
    """qproj_attention against fp32 torch: LayerNorm -> Linear -> softmax(q k^T / 8) v per head.  qcfg forces the tile (1: 256 x 256,
    2: 128 x 128 with mover waves); None = the launch's own choice."""
    if qcfg:
        ENV["CD360_QATTN_CFG"] = str(qcfg)
    try:
        return _check_qattn(b, nq, C, K, nk, ln, qcfg)
    finally:
        ENV.pop("CD360_QATTN_CFG", None)


def _check_qattn(b, nq, C, K, nk, ln, qcfg):
    heads = C // 64
    a = (rnd(b, nq, K, seed=11) * (0.5 + rnd(b, nq, 1, seed=12).abs()) + 0.5 * rnd(b, nq, 1, seed=13)).to(torch.bfloat16)
    w = rnd(C, K, seed=14, scale=K ** -0.5)
    kv = rnd(b, max(80, nk), 2 * C, seed=15).to(torch.bfloat16)
    k, v = kv[..., :C], kv[..., C:]
    a32 = a.float()
    if ln:
        gamma, beta = 1 + 0.2 * rnd(K, seed=5), 0.1 * rnd(K, seed=6)
        wp, wsum, cb = ops.pack_ln_linear(w, None, gamma, beta)
        q = F.linear(F.layer_norm(a32, (K,), gamma, beta, 1e-5), w)
        got = [ops.qproj_attention(a, wp, k, v, nk, heads, bias=cb, ln=(ops.row_stats(a), wsum, 1e-5)) for _ in range(3)]
    else:
        wb = w.to(torch.bfloat16)
        q = F.linear(a32, wb.float())
        got = [ops.qproj_attention(a, wb, k, v, nk, heads) for _ in range(3)]
    qh = q.reshape(b, nq, heads, 64).transpose(1, 2)
    kh = k[:, :nk].float().reshape(b, nk, heads, 64).transpose(1, 2)
    vh = v[:, :nk].float().reshape(b, nk, heads, 64).transpose(1, 2)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(b, nq, C)
    torch.cuda.synchronize()
    e = relerr(got[0], want)
    same = all(torch.equal(got[0], g) for g in got[1:])
    ok = same and e < 1e-2 and bool(torch.isfinite(got[0]).all())
    print(("ok   " if ok else "FAIL ") + f"qproj_attention b={b} nq={nq} C={C} K={K} nk={nk} ln={int(ln)} tile={qcfg or 'auto'}: err {e:.2e} repeat-equal {same}", flush=True)
    return ok


def check():
    ok = check_qattn_all()
    for cfg in (1, 2, 3, 4, 5, 6, 7, 8):
        for (M, N, K) in [(256, 256, 64), (256, 256, 128), (512, 512, 192), (300, 272, 320), (128, 128, 64), (1000, 640, 640), (3072, 1280, 1280)]:
            ok &= check_case(M, N, K, cfg=cfg)
        ok &= check_case(520, 640, 320, bias=True, res=True, stats=True, cfg=cfg)
        ok &= check_case(777, 1280, 640, bias=True, ln=True, cfg=cfg)
        ok &= check_case(1024, 384, 256, bias=True, ln=True, res=True, stats=True, cfg=cfg, lda=512)
    ok &= check_case(515, 1280, 320, bias=True, ln=True, geglu=True)
    for cfg in (3, 5, 7):  # GEGLU on every tiling that has value / gate block pairs
        ok &= check_case(515, 1280, 320, bias=True, ln=True, geglu=True, cfg=cfg)
        ok &= check_case(777, 640, 192, bias=True, geglu=True, cfg=cfg)
    ok &= check_case(3072, 10240, 1280, bias=True, ln=True, geglu=True)
    ok &= check_case(12288, 1920, 640, bias=True, ln=True)
    ok &= check_case(12288, 640, 2560, bias=True, res=True, stats=True)
    ok &= check_case(3072, 1280, 5120, bias=True, res=True, stats=True)
    for (M, N, K) in ((1024, 1280, 5120), (1000, 1280, 10240), (320, 2560, 2048), (1024, 5120, 1280)):  # small batches: 64 x 128 tiles (both wave
        ok &= check_case(M, N, K, bias=True, res=True, stats=True)                                    # arrangements), 256 x 128 with three buffers
        ok &= check_case(M, N, K, bias=True, ln=True)
    print("CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def timeit(fn, iters=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best * 1e3  # us


def timeit_graph(fn, n=50, reps=5):
    """GPU-side time per call: n calls captured in one hipGraph and replayed (the Python / ctypes launch path costs ~15 us per call,
    more than the short kernels run -- `timeit` on those measures the host)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


SHAPES = [  # (name, M, N, K, epilogue) at cfg-B: b = 3; level 1: 4096 tokens x 640, level 2: 1024 tokens x 1280
    ("L1 qkv", 12288, 1920, 640, "ln"), ("L1 out", 12288, 640, 640, "res"), ("L1 ff1", 12288, 5120, 640, "geglu"), ("L1 ff2", 12288, 640, 2560, "res"),
    ("L2 qkv", 3072, 3840, 1280, "ln"), ("L2 out", 3072, 1280, 1280, "res"), ("L2 ff1", 3072, 10240, 1280, "geglu"), ("L2 ff2", 3072, 1280, 5120, "res"),
    ("A3 q L1", 98304 * 3, 640, 640, "ln"), ("A3 q L2", 24576 * 3, 1280, 1280, "ln"), ("4k cube", 4096, 4096, 4096, ""),
]


VARIANTS = [(0, 0)]  # 0 = the tiling pick_cfg chooses (the product path); CD360_GEMM_CFG=n forces one


def time_qattn():
    """The pose-token attention (A3; plain b = 3 and the de-duplicated CFG form the render step runs) and the text cross-attention (A2) at
    cfg-B: fused kernel (auto tile, every forced tile, and the A/B partners of the round-4 epilogue: keys padded to 96, no column split)
    against q GEMM + small-Nk attention; hipGraph-timed.  Run it under CD360_LIB=<another build> for a same-box A/B of two libraries."""
    from cd360 import _lib
    print("library:", _lib.LIB_PATH, flush=True)
    for name, b, nq, C, dup in (("A3 L1", 3, 98304, 640, 0), ("A3 L2", 3, 24576, 1280, 0), ("A3 L1 dedup", 2, 98304, 640, 1), ("A3 L2 dedup", 2, 24576, 1280, 1),
                                ("A2 L1", 3, 4096, 640, 0), ("A2 L2", 3, 1024, 1280, 0)):
        heads = C // 64
        a = rnd(b, nq, C, seed=1).to(torch.bfloat16)
        w = rnd(C, C, seed=2, scale=C ** -0.5).to(torch.bfloat16)
        ws = w.float().sum(1).contiguous()
        cb = rnd(C, seed=3)
        kv = rnd(b + dup, 80, 2 * C, seed=4).to(torch.bfloat16)
        k, v = kv[..., :C], kv[..., C:]
        st = ops.row_stats(a)
        n = 10 if nq > 50000 else 30
        run = lambda: ops.qproj_attention(a, w, k, v, 77, heads, bias=cb, ln=(st, ws, 1e-5), dup=dup)
        flops = (b * nq * 2.0 * C * C + (b + dup) * nq * 4.0 * 77 * C)
        line = [f"{name}: auto {timeit_graph(run, n=n):7.1f} us"]
        line[0] += f" ({flops / float(line[0].split()[-2]) * 1e-6:5.0f} TF/s)"
        for field in ("qattn_keys16", "qattn_split"):
            try:
                with _lib.tuning(**{field: 0}):
                    line.append(f"{field}=0 {timeit_graph(run, n=n):7.1f}")
            except Exception as e:  # an older library build
                line.append(f"{field}=0 n/a")
        for qcfg in (1, 2, 3, 4):
            with _lib.tuning(qattn_cfg=qcfg):
                line.append(f"tile{qcfg} {timeit_graph(run, n=n):7.1f}")
        if not dup:
            t_g = timeit_graph(lambda: ops.gemm(a, w, bias=cb, ln=(st, ws, 1e-5)), n=n)
            q = ops.gemm(a, w, bias=cb, ln=(st, ws, 1e-5))
            t_a = timeit_graph(lambda: ops.attention(q, k, v, heads, 77), n=n)
            line.append(f"q GEMM {t_g:7.1f} + attention {t_a:7.1f}")
        print(" | ".join(line), flush=True)


def check_qattn_fp8(b, nq, C, K, nk, qcfg=None, dup=0):
    """cd360_qproj_attn_fp8_bf16 (BASELINE configs[4]: q K^T and P V on fp8 MFMA) against fp32 torch and against the bf16 kernel on the same
    inputs.  e4m3 carries three mantissa bits (2^-4 per value, and the logits feel it: measured 6e-2 ... 1e-1 of the output's max magnitude on
    unit-scale inputs, against 4e-3 for bf16): the bar here only guards against layout / scale bugs (1.2e-1), the tolerance REPORT is
    bench.py --fp8-attn's; returns
    (ok, error vs fp32, error of the bf16 kernel vs fp32)."""
    from cd360 import _lib
    heads = C // 64
    a = (rnd(b, nq, K, seed=11) * (0.5 + rnd(b, nq, 1, seed=12).abs()) + 0.5 * rnd(b, nq, 1, seed=13)).to(torch.bfloat16)
    w = rnd(C, K, seed=14, scale=K ** -0.5)
    kv = rnd(b + dup, max(80, nk), 2 * C, seed=15).to(torch.bfloat16)
    k, v = kv[..., :C], kv[..., C:]
    gamma, beta = 1 + 0.2 * rnd(K, seed=5), 0.1 * rnd(K, seed=6)
    wp, wsum, cb = ops.pack_ln_linear(w, None, gamma, beta)
    packed = ops.kv_pack_fp8(k, v, nk, heads)
    with _lib.tuning(qattn_cfg=qcfg or -1):
        got = [ops.qproj_attention(a, wp, k, v, nk, heads, bias=cb, ln=(ops.row_stats(a), wsum, 1e-5), dup=dup, fp8=packed) for _ in range(3)]
        ref16 = ops.qproj_attention(a, wp, k, v, nk, heads, bias=cb, ln=(ops.row_stats(a), wsum, 1e-5), dup=dup)
    a3 = torch.cat([a, a[b - dup:]], 0) if dup else a
    q = F.linear(F.layer_norm(a3.float(), (K,), gamma, beta, 1e-5), w)
    qh = q.reshape(b + dup, nq, heads, 64).transpose(1, 2)
    kh = k[:, :nk].float().reshape(b + dup, nk, heads, 64).transpose(1, 2)
    vh = v[:, :nk].float().reshape(b + dup, nk, heads, 64).transpose(1, 2)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(1, 2).reshape(b + dup, nq, C)
    torch.cuda.synchronize()
    e8, e16 = relerr(got[0], want), relerr(ref16, want)
    same = all(torch.equal(got[0], g) for g in got[1:])
    ok = same and e8 < 1.2e-1 and bool(torch.isfinite(got[0]).all())
    print(("ok   " if ok else "FAIL ") + f"qproj_attention fp8 b={b} nq={nq} C={C} K={K} nk={nk} dup={dup} tile={qcfg or 'auto'}: err {e8:.2e} "
          f"(bf16 kernel {e16:.2e}) repeat-equal {same}", flush=True)
    return ok, e8, e16


def check_qattn_fp8_all():
    ok = True
    for qcfg in (1, 2, 3, 4):
        ok &= check_qattn_fp8(2, 512, 640, 640, 77, qcfg=qcfg)[0]
        ok &= check_qattn_fp8(1, 256, 192, 128, 65, qcfg=qcfg)[0]
    ok &= check_qattn_fp8(3, 1024, 1280, 1280, 77)[0]
    ok &= check_qattn_fp8(1, 256, 128, 64, 96)[0]
    ok &= check_qattn_fp8(2, 256, 128, 128, 80, dup=1)[0]
    ok &= check_qattn_fp8(2, 1024, 640, 640, 77, dup=1, qcfg=1)[0]
    # the de-duplicated form equals the expanded batch bit for bit, as the bf16 kernel's does
    b, dup, nq, C, nk = 2, 1, 512, 640, 77
    a = (rnd(b, nq, C, seed=21) * (0.5 + rnd(b, nq, 1, seed=22).abs())).to(torch.bfloat16)
    w = rnd(C, C, seed=24, scale=C ** -0.5).to(torch.bfloat16)
    kv = rnd(b + dup, 80, 2 * C, seed=25).to(torch.bfloat16)
    packed = ops.kv_pack_fp8(kv[..., :C], kv[..., C:], nk, C // 64)
    got = ops.qproj_attention(a, w, kv[..., :C], kv[..., C:], nk, C // 64, dup=dup, fp8=packed)
    a3 = torch.cat([a, a[b - dup:]], 0).contiguous()
    want = ops.qproj_attention(a3, w, kv[..., :C], kv[..., C:], nk, C // 64, fp8=packed)
    same = torch.equal(got, want)
    print(("ok   " if same else "FAIL ") + "fp8 de-duplicated == expanded batch", flush=True)
    ok &= same
    print("QATTN FP8 CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def check_qattn_all():
    """Parity list of the fused query projection + attention alone (every tile, every key-count epilogue, one-tile K loops, the column split)."""
    ok = True
    ok &= check_qattn(2, 256, 128, 128, 77)
    ok &= check_qattn(1, 512, 64, 64, 77, ln=False)
    ok &= check_qattn(3, 1024, 640, 640, 77)
    ok &= check_qattn(2, 256, 1280, 1280, 50)
    ok &= check_qattn(1, 768, 256, 320, 96)
    ok &= check_qattn(2, 256, 128, 192, 20)
    ok &= check_qattn(1, 256, 384, 256, 77)   # 384 = 256 + 128 columns: the split launch
    ok &= check_qattn(1, 512, 640, 64, 80)    # one K-tile
    ok &= check_qattn(1, 256, 640, 128, 65)   # two K-tiles
    ok &= check_qattn(1, 256, 320, 192, 81)   # three K-tiles, 96-key epilogue
    for qcfg in (1, 2, 3, 4):  # every tile on the same shapes; 128-token multiples only on the 128-row tiles
        ok &= check_qattn(3, 1024, 1280, 1280, 77, qcfg=qcfg)
        ok &= check_qattn(2, 512, 640, 640, 77, qcfg=qcfg)
        ok &= check_qattn(1, 256, 192, 128, 33, qcfg=qcfg)
        ok &= check_qattn(1, 256, 128, 64, 77, qcfg=qcfg)
        ok &= check_qattn(1, 256, 256, 320, 16, qcfg=qcfg)
    for qcfg in (2, 3):
        ok &= check_qattn(2, 384, 320, 320, 77, qcfg=qcfg)
        ok &= check_qattn(3, 128, 128, 64, 20, ln=False, qcfg=qcfg)
    from cd360 import _lib
    with _lib.tuning(qattn_keys16=0, qattn_split=0):
        ok &= check_qattn(3, 1024, 640, 640, 77)
        ok &= check_qattn(2, 256, 1280, 1280, 77, qcfg=2)
    print("QATTN CHECK", "PASSED" if ok else "FAILED", flush=True)
    return ok


def time_all():
    for name, M, N, K, epi in SHAPES:
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        b16 = b32.to(torch.bfloat16)
        r = rnd(M, N, seed=4).to(torch.bfloat16) if epi == "res" else None
        flops = 2.0 * M * N * K
        t_lib = timeit_graph(lambda: F.linear(a, w, b16), n=20)
        line = f"{name:8s} M={M:6d} N={N:5d} K={K:4d} | hipBLASLt {t_lib:7.1f} us {flops / t_lib * 1e-6:6.0f} TF"
        for cfg, sched in VARIANTS:
            if epi == "geglu" and cfg in (2, 4, 6):
                continue
            if cfg:
                ENV["CD360_GEMM_CFG"] = str(cfg)
            t_plain = timeit_graph(lambda: ops.gemm(a, w, bias=b32), n=20)
            line += f" | cd360 {'auto' if not cfg else 'cfg%d' % cfg} plain {t_plain:7.1f} us {flops / t_plain * 1e-6:6.0f} TF"
            if epi == "ln":
                st = ops.row_stats(a)
                ws = w.float().sum(1).contiguous()
                t = timeit_graph(lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5)), n=20)
                line += f" ln {t:7.1f}"
            elif epi == "res":
                t = timeit_graph(lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True), n=20)
                line += f" res+stats {t:7.1f}"
            elif epi == "geglu":
                st = ops.row_stats(a)
                ws = w.float().sum(1).contiguous()
                t = timeit_graph(lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=True), n=20)
                line += f" ln+geglu {t:7.1f}"
        if epi == "geglu":  # what the fused call replaces: library GEMM + geglu pass (+ the LayerNorm pass in front, not timed here)
            t2 = timeit_graph(lambda: ops.geglu(F.linear(a, w, b16)), n=20)
            line += f" | lib+geglu {t2:7.1f}"
        print(line, flush=True)
    ENV.pop("CD360_GEMM_CFG", None)
    ENV.pop("CD360_GEMM_SCHED", None)


def ablate():
    """What-if timings (CD360_GEMM_ABL; results are garbage by construction): what the waits, the DMA and the stores cost."""
    for (M, N, K) in ((4096, 4096, 4096), (3072, 10240, 1280), (3072, 3840, 1280), (12288, 5120, 640)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        print(f"M={M} N={N} K={K}")
        for cfg in (3, 6, 1):
            ENV["CD360_GEMM_CFG"] = str(cfg)
            line = f"cfg{cfg}:"
            for abl, name in ((0, "full"), (64, "no stores"), (16, "no barrier"), (56, "no waits at all"), (4, "no DMA"), (60, "no DMA no waits"), (124, "no DMA/waits/stores")):
                ENV["CD360_GEMM_ABL"] = str(abl)
                t = timeit(lambda: ops.gemm(a, w), iters=20, warm=3)
                line += f" | {name} {t:6.1f}"
            print(line, flush=True)
    for k in ("CD360_GEMM_ABL", "CD360_GEMM_CFG"):
        ENV.pop(k, None)


def group_m_sweep():
    """Tile-group height (CD360_GEMM_GROUP_M: token tiles per group, the channel tile varying slowest inside a group) on the block's shapes."""
    for name, M, N, K, geglu in (("L2 ff1", 3072, 10240, 1280, True), ("L2 qkv", 3072, 3840, 1280, False), ("L2 ff2", 3072, 1280, 5120, False),
                                 ("L1 ff1", 12288, 5120, 640, True), ("L1 qkv", 12288, 1920, 640, False), ("L2 out", 3072, 1280, 1280, False)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        st, ws = ops.row_stats(a), w.float().sum(1).contiguous()
        line = f"{name} M={M} N={N} K={K}:"
        for gm in (1, 2, 3, 4, 6, 12, 48):
            ENV["CD360_GEMM_GROUP_M"] = str(gm)
            t = timeit(lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=geglu))
            line += f" | gm{gm} {t:6.1f}"
        print(line, flush=True)
    ENV.pop("CD360_GEMM_GROUP_M", None)


def ablate_small():
    """The same what-if timings for the narrow-output shapes of the 1280 level (128 x 128 tilings, one workgroup per CU)."""
    for (M, N, K) in ((3072, 1280, 1280), (3072, 1280, 5120), (12288, 640, 2560)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        z = torch.zeros_like(a), torch.zeros_like(w)
        print(f"M={M} N={N} K={K}")
        for cfg in (4, 2, 1, 5):
            ENV["CD360_GEMM_CFG"] = str(cfg)
            line = f"cfg{cfg}:"
            for abl, name in ((0, "full"), (64, "no stores"), (16, "no barrier"), (8, "no DMA wait"), (32, "no LDS wait"), (56, "no waits"), (4, "no DMA"),
                              (60, "no DMA no waits"), (124, "no DMA/waits/stores")):
                ENV["CD360_GEMM_ABL"] = str(abl)
                t = timeit(lambda: ops.gemm(a, w), iters=20, warm=3)
                line += f" | {name} {t:6.1f}"
            ENV["CD360_GEMM_ABL"] = "0"
            t = timeit(lambda: ops.gemm(z[0], z[1]), iters=20, warm=3)
            line += f" | zeros {t:6.1f}"
            print(line, flush=True)
    for k in ("CD360_GEMM_ABL", "CD360_GEMM_CFG"):
        ENV.pop(k, None)


def ksplit(time=True):
    """The 128 x 128 / four-buffer tiling in its three wave arrangements (CD360_GEMM_KSPLIT: 0 = eight waves of 64 x 32, 1 = eight waves of
    64 x 64 in two k-step groups, 2 = four waves of 64 x 64): parity on ragged shapes and every epilogue, then timing on the narrow shapes
    and the 32^2 convolutions."""
    ok = True
    for mode in (1, 2):
        ENV["CD360_GEMM_KSPLIT"] = str(mode)
        for (M, N, K) in [(256, 256, 64), (256, 256, 128), (512, 512, 192), (300, 272, 320), (128, 128, 64), (1000, 640, 640), (3072, 1280, 1280)]:
            ok &= check_case(M, N, K, cfg=4)
        ok &= check_case(520, 640, 320, bias=True, res=True, stats=True, cfg=4)
        ok &= check_case(777, 1280, 640, bias=True, ln=True, cfg=4)
        ok &= check_case(1024, 384, 256, bias=True, ln=True, res=True, stats=True, cfg=4, lda=512)
        ok &= check_case(3072, 1280, 5120, bias=True, res=True, stats=True, cfg=4)
    print("KSPLIT CHECK", "PASSED" if ok else "FAILED", flush=True)
    # timing: the arrangements interleaved over three rounds, best of each (a first-timed variant otherwise pays the clock ramp)
    for name, M, N, K in (("L2 out", 3072, 1280, 1280), ("L2 ff2", 3072, 1280, 5120), ("L2 pose", 3072, 1280, 2560)) if time else ():
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        r = rnd(M, N, seed=4).to(torch.bfloat16)
        ENV["CD360_GEMM_CFG"] = "4"
        best = {0: 1e9, 1: 1e9, 2: 1e9}
        for _ in range(3):
            for mode in (0, 1, 2):
                ENV["CD360_GEMM_KSPLIT"] = str(mode)
                best[mode] = min(best[mode], timeit(lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)))
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:4d} | " + " | ".join(f"mode{m} {t:6.1f}" for m, t in best.items()), flush=True)
    ENV.pop("CD360_GEMM_CFG", None)
    for (N_, H, W, cin, cout) in ((3, 32, 32, 1280, 1280), (3, 32, 32, 2560, 1280), (3, 32, 32, 1920, 1280)):
        x = torch.randn(N_, H * W, cin, device=dev).to(torch.bfloat16)
        wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
        bias = torch.randn(cout, device=dev)
        ENV["CD360_CONV_CFG"] = "4"
        outs, best = {}, {0: 1e9, 1: 1e9, 2: 1e9}
        for _ in range(3 if time else 1):
            for mode in (0, 1, 2):
                ENV["CD360_GEMM_KSPLIT"] = str(mode)
                outs[mode] = ops.conv_igemm(x, wp, bias, N_, H, W, 9, want_stats=True)
                if time:
                    best[mode] = min(best[mode], timeit(lambda: ops.conv_igemm(x, wp, bias, N_, H, W, 9)))
        line = f"conv {N_}x{H}x{W} {cin}->{cout}: " + " | ".join(f"mode{m} {t:6.1f}" for m, t in best.items())
        y0, st0 = outs[0][0].float(), outs[0][1]
        for mode in (1, 2):
            y, st = outs[mode]
            e = ((y.float() - y0).abs().max() / y0.abs().max()).item()
            es = ((st - st0).abs().max() / st0.abs().max()).item()
            ok &= e < 1e-2 and es < 1e-3
            line += f" | mode{mode} err {e:.1e} stats {es:.1e}"
        print(line, flush=True)
    for k in ("CD360_CONV_CFG", "CD360_GEMM_KSPLIT"):
        ENV.pop(k, None)
    return ok


def narrow():
    """The short launches of the block (auto tiling, bias + residual + row statistics) and the 3 x 3 convolutions: best of three rounds."""
    line = []
    for name, M, N, K in (("L2 out", 3072, 1280, 1280), ("L2 ff2", 3072, 1280, 5120), ("L1 out", 12288, 640, 640), ("L1 ff2", 12288, 640, 2560)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32, r = rnd(N, seed=3), rnd(M, N, seed=4).to(torch.bfloat16)
        line.append(f"{name} {timeit_graph(lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)):6.1f}")
    for (N_, H, W, cin, cout) in ((3, 32, 32, 1280, 1280), (3, 64, 64, 640, 640), (3, 128, 128, 320, 320)):
        x = torch.randn(N_, H * W, cin, device=dev).to(torch.bfloat16)
        wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
        bias, emb = torch.randn(cout, device=dev), torch.randn(N_, cout, device=dev).to(torch.bfloat16)
        line.append(f"conv{H} {timeit_graph(lambda: ops.conv_igemm(x, wp, bias, N_, H, W, 9, emb=emb, want_stats=True)):6.1f}")
    print(" | ".join(line), flush=True)


def small_m():
    """The fine-tune step's target stream at the 1280 level (batch 4 x 256 tokens = 1024 rows; backward dX shapes included): 128 x 128 tiles
    (tiling 4: 80 workgroups) against 64 x 128 tiles (tiling 8: 160), bias + residual epilogue; parity of tiling 8 against fp32 first."""
    for (M, N, K) in ((1024, 1280, 1280), (1024, 1280, 3840), (1024, 1280, 5120), (1024, 1280, 10240), (1024, 3840, 1280), (1024, 5120, 1280),
                      (1024, 10240, 1280), (768, 1280, 1280), (320, 2560, 2048), (1280, 2560, 2048), (4096, 640, 640), (4096, 1280, 1280), (1000, 1280, 1280)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32, r = rnd(N, seed=3), rnd(M, N, seed=4).to(torch.bfloat16)
        want = F.linear(a.float(), w.float(), b32) + r.float()
        line = f"M={M:5d} N={N:5d} K={K:5d}:"
        for cfg in (0, 4, 8):
            if cfg:
                ENV["CD360_GEMM_CFG"] = str(cfg)
            else:
                ENV.pop("CD360_GEMM_CFG", None)
            got, st = ops.gemm(a, w, bias=b32, res=r, want_stats=True)
            err = relerr(got, want)
            serr = (st.sum(1)[:, 0] - got.float().sum(1)).abs().max().item()
            assert err < 2e-2 and serr < 0.5, (cfg, M, N, K, err, serr)
            line += f"  cfg{cfg} {timeit_graph(lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)):6.1f} us (err {err:.1e})"
        ENV.pop("CD360_GEMM_CFG", None)
        print(line, flush=True)


def mid_m():
    """Wide outputs of a small batch (fine-tune step: merged q|k|v, context k|v, GEGLU backward): every linear tiling."""
    shapes = ((1024, 3840, 1280), (1024, 5120, 1280), (1024, 10240, 1280), (1280, 2560, 2048), (320, 2560, 2048), (320, 1280, 2048),
              (4096, 3840, 1280), (4096, 5120, 640), (4096, 1920, 640), (2048, 1280, 1280), (1536, 1280, 1280), (3072, 3840, 1280))
    if os.environ.get("CD360_BENCH_SHAPES") == "cfgB":  # the linear launches of the sampling step (b = 3 at 1024^2)
        shapes = ((12288, 640, 640), (12288, 640, 2560), (12288, 1920, 640), (3072, 1280, 1280), (3072, 1280, 5120), (3072, 3840, 1280),
                  (231, 2560, 2048), (231, 1280, 2048), (3072, 1280, 2560), (12288, 640, 1280), (49152, 320, 320))
    if os.environ.get("CD360_BENCH_SHAPES") == "narrow":  # narrow outputs with 257 .. 640 tiles of 128 x 128: two-buffer 128 x 128 or 256 x 128?
        shapes = ((4096, 1280, 1280), (4096, 1280, 5120), (4096, 1280, 3840), (16384, 640, 640), (12288, 640, 640), (12288, 640, 2560), (5120, 1280, 1280),
                  (6144, 1280, 1280), (8192, 640, 640), (8192, 640, 2560))
    for (M, N, K) in shapes:
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        line = f"M={M:5d} N={N:5d} K={K:5d}:"
        for cfg in (0, 1, 2, 3, 4, 5, 6, 8):
            if cfg:
                ENV["CD360_GEMM_CFG"] = str(cfg)
            else:
                ENV.pop("CD360_GEMM_CFG", None)
            line += f"  cfg{cfg} {timeit_graph(lambda: ops.gemm(a, w, bias=b32)):6.1f}"
        ENV.pop("CD360_GEMM_CFG", None)
        print(line, flush=True)


def fixed_cost():
    """Launch time against K on the 1280-level C -> C shape (M = 3072, N = 1280): the intercept is what a launch costs besides its K loop."""
    M, N = 3072, 1280
    b32, r = rnd(N, seed=3), rnd(M, N, seed=4).to(torch.bfloat16)
    for K in (64, 128, 256, 512, 1280, 2560):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        line = f"K={K:5d}: full epilogue {timeit_graph(lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)):6.1f}"
        line += f" | plain {timeit_graph(lambda: ops.gemm(a, w)):6.1f}"
        ENV["CD360_GEMM_ABL"] = "64"
        line += f" | plain, no stores {timeit_graph(lambda: ops.gemm(a, w)):6.1f}"
        ENV.pop("CD360_GEMM_ABL")
        print(line, flush=True)
    x = torch.zeros(64, device=dev)
    print(f"torch elementwise on 64 floats (launch floor in a graph): {timeit_graph(lambda: x.add_(1.0)):6.1f}", flush=True)


def movers(time=True):
    """Dedicated mover waves (CD360_GEMM_MOVERS=4: four extra waves issue all LDS-DMA pieces, the eight others only multiply) on the
    128 x 128 four-buffer tiling, unsplit and with the k-step groups: bit-identical results, hipGraph-timed, interleaved."""
    ok = True
    variants = (("ks0", "0", "0"), ("ks0+mv", "0", "4"), ("ks1", "1", "0"), ("ks1+mv", "1", "4"))
    for name, M, N, K in (("L2 out", 3072, 1280, 1280), ("L2 pose", 3072, 1280, 2560), ("L2 ff2", 3072, 1280, 5120), ("ragged", 1000, 640, 640), ("tiny", 300, 272, 320)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32, r = rnd(N, seed=3), rnd(M, N, seed=4).to(torch.bfloat16)
        ENV["CD360_GEMM_CFG"] = "4"
        best, outs = {v[0]: 1e9 for v in variants}, {}
        for _ in range(3):
            for tag, ks, mv in variants:
                ENV["CD360_GEMM_KSPLIT"], ENV["CD360_GEMM_MOVERS"] = ks, mv
                outs[tag] = ops.gemm(a, w, bias=b32, res=r, want_stats=True)
                if time:
                    best[tag] = min(best[tag], timeit_graph(lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)))
        same = all(torch.equal(outs[x][i], outs[x + "+mv"][i]) for x in ("ks0", "ks1") for i in (0, 1))
        ok &= same
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:4d} | " + " | ".join(f"{k} {v:6.1f}" for k, v in best.items()) + f" | movers == plain: {same}", flush=True)
    ENV.pop("CD360_GEMM_CFG", None)
    for (N_, H, W, cin, cout) in ((3, 32, 32, 1280, 1280), (3, 32, 32, 2560, 1280), (2, 16, 16, 128, 192)):
        x = torch.randn(N_, H * W, cin, device=dev).to(torch.bfloat16)
        wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
        bias, emb = torch.randn(cout, device=dev), torch.randn(N_, cout, device=dev).to(torch.bfloat16)
        ENV["CD360_CONV_CFG"] = "4"
        best, outs = {v[0]: 1e9 for v in variants}, {}
        for _ in range(3):
            for tag, ks, mv in variants:
                ENV["CD360_GEMM_KSPLIT"], ENV["CD360_GEMM_MOVERS"] = ks, mv
                outs[tag] = ops.conv_igemm(x, wp, bias, N_, H, W, 9, emb=emb, want_stats=True)
                if time:
                    best[tag] = min(best[tag], timeit_graph(lambda: ops.conv_igemm(x, wp, bias, N_, H, W, 9, emb=emb, want_stats=True)))
        same = all(torch.equal(outs[x_][i], outs[x_ + "+mv"][i]) for x_ in ("ks0", "ks1") for i in (0, 1))
        ok &= same
        print(f"conv {N_}x{H}x{W} {cin}->{cout}: " + " | ".join(f"{k} {v:6.1f}" for k, v in best.items()) + f" | movers == plain: {same}", flush=True)
    for k in ("CD360_CONV_CFG", "CD360_GEMM_KSPLIT", "CD360_GEMM_MOVERS"):
        ENV.pop(k, None)
    print("MOVERS", "PASSED" if ok else "FAILED", flush=True)
    return ok


def movers_all():
    """CD360_GEMM_MOVERS = 0 against 4 on the shapes of the other tilings (auto tiling), hipGraph-timed, interleaved, bit-equality checked."""
    ok = True
    for name, M, N, K, epi in (("L1 out", 12288, 640, 640, "res"), ("L1 ff2", 12288, 640, 2560, "res"), ("L1 pose", 12288, 640, 1280, "res"),
                               ("L2 qkv", 3072, 3840, 1280, "ln"), ("L1 qkv", 12288, 1920, 640, "ln"), ("L2 out", 3072, 1280, 1280, "res"), ("L2 ff2", 3072, 1280, 5120, "res")):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32, r = rnd(N, seed=3), rnd(M, N, seed=4).to(torch.bfloat16)
        st, ws = ops.row_stats(a), w.float().sum(1).contiguous()
        fn = (lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)) if epi == "res" else (lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5)))
        best, outs = {"0": 1e9, "4": 1e9}, {}
        for _ in range(3):
            for mv in ("0", "4"):
                ENV["CD360_GEMM_MOVERS"] = mv
                outs[mv] = fn()
                best[mv] = min(best[mv], timeit_graph(fn))
        o0, o4 = (outs[k] if isinstance(outs[k], tuple) else (outs[k],) for k in ("0", "4"))
        same = all(torch.equal(x, y) for x, y in zip(o0, o4))
        ok &= same
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:4d} tile_n={ops._lib.load().cd360_gemm_tile_n(M, N)} | movers 0: {best['0']:6.1f} | 4: {best['4']:6.1f} | equal: {same}", flush=True)
    for (N_, H, W, cin, cout) in ((3, 64, 64, 640, 640), (3, 64, 64, 1280, 640), (3, 64, 64, 1920, 640), (3, 128, 128, 320, 320), (3, 32, 32, 1280, 1280)):
        x = torch.randn(N_, H * W, cin, device=dev).to(torch.bfloat16)
        wp = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).to(torch.bfloat16)
        bias, emb = torch.randn(cout, device=dev), torch.randn(N_, cout, device=dev).to(torch.bfloat16)
        fn = lambda: ops.conv_igemm(x, wp, bias, N_, H, W, 9, emb=emb, want_stats=True)
        best, outs = {"0": 1e9, "4": 1e9}, {}
        for _ in range(3):
            for mv in ("0", "4"):
                ENV["CD360_GEMM_MOVERS"] = mv
                outs[mv] = fn()
                best[mv] = min(best[mv], timeit_graph(fn))
        same = torch.equal(outs["0"][0], outs["4"][0]) and torch.equal(outs["0"][1], outs["4"][1])
        ok &= same
        print(f"conv {N_}x{H}x{W} {cin}->{cout}: movers 0: {best['0']:6.1f} | 4: {best['4']:6.1f} | equal: {same}", flush=True)
    ENV.pop("CD360_GEMM_MOVERS", None)
    print("MOVERS_ALL", "PASSED" if ok else "FAILED", flush=True)
    return ok


def ff1_tilings():
    """FF1 + GEGLU: 256 x 256 tiles (no room for mover waves) against 256 x 128 tiles with and without them."""
    for name, M, N, K in (("L2 ff1", 3072, 10240, 1280), ("L1 ff1", 12288, 5120, 640)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        st, ws = ops.row_stats(a), w.float().sum(1).contiguous()
        fn = lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=True)
        variants = (("256x256", "3", "0"), ("256x256/16w", "7", "0"), ("256x128", "5", "0"), ("256x128+movers", "5", "4"))
        best, outs = {v[0]: 1e9 for v in variants}, {}
        for _ in range(3):
            for tag, cfg, mv in variants:
                ENV["CD360_GEMM_CFG"], ENV["CD360_GEMM_MOVERS"] = cfg, mv
                outs[tag] = fn()
                best[tag] = min(best[tag], timeit_graph(fn, n=20))
        same = torch.equal(outs["256x128"], outs["256x128+movers"])
        e16 = relerr(outs["256x256/16w"], outs["256x256"].float())
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:4d} | " + " | ".join(f"{k} {v:6.1f}" for k, v in best.items()) + f" | movers == plain: {same} | 16w vs 8w err {e16:.1e}", flush=True)
    for k in ("CD360_GEMM_CFG", "CD360_GEMM_MOVERS"):
        ENV.pop(k, None)


def geglu_cost():
    """What the GEGLU epilogue (erf GELU of the gate, product with the value) costs on FF1: the same launch with and without it
    (same tiling, same LayerNorm fold and bias)."""
    for name, M, N, K in (("L2 ff1", 3072, 10240, 1280), ("L1 ff1", 12288, 5120, 640)):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        st, ws = ops.row_stats(a), w.float().sum(1).contiguous()
        line = f"{name}:"
        for cfg in ("7", "3"):
            ENV["CD360_GEMM_CFG"] = cfg
            g = timeit_graph(lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=True), n=20)
            p_ = timeit_graph(lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5)), n=20)
            line += f"  cfg{cfg}: GEGLU {g:6.1f} us, plain (2x the output bytes) {p_:6.1f} us"
        ENV.pop("CD360_GEMM_CFG", None)
        print(line, flush=True)


def whatif():
    """hipGraph-timed what-if builds of the 128 x 128 four-buffer tiling on the long-K shape (results invalid by construction)."""
    M, N = 3072, 1280
    for K in (1280, 5120):
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        for ks in ("0", "1"):
            ENV["CD360_GEMM_KSPLIT"] = ks
            line = f"K={K} ksplit={ks}:"
            for abl, name in ((0, "full"), (64, "no stores"), (512, "DMA + rendezvous only"), (512 + 16, "DMA only, no barrier"), (512 + 16 + 8, "DMA issue only, no waits"),
                              (4 + 64, "no DMA, no stores"), (4 + 8 + 16 + 32 + 64, "reads + MFMAs only")):
                ENV["CD360_GEMM_ABL"] = str(abl)
                line += f" | {name} {timeit_graph(lambda: ops.gemm(a, w)):6.1f}"
            print(line, flush=True)
    for k in ("CD360_GEMM_ABL", "CD360_GEMM_KSPLIT"):
        ENV.pop(k, None)


def stride_sweep():
    """Does the row stride of the operands matter (rows of a K-tile 2560 / 10240 bytes apart fall on few L2 channels)?  Same GEMM with the
    A and W rows padded by 0 / 64 / 32 / 8 elements."""
    for name, M, N, K in (("L2 out", 3072, 1280, 1280), ("L2 ff2", 3072, 1280, 5120), ("L1 out", 12288, 640, 640), ("L2 qkv", 3072, 3840, 1280)):
        line = f"{name:8s} M={M:6d} N={N:5d} K={K:4d}:"
        for pad_a, pad_w in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (8, 8), (192, 192)):
            a = rnd(M, K + pad_a, seed=1).to(torch.bfloat16)[:, :K]
            w = rnd(N, K + pad_w, seed=2, scale=K ** -0.5).to(torch.bfloat16)[:, :K]
            b32 = rnd(N, seed=3)
            line += f" | a+{pad_a} w+{pad_w}: {timeit_graph(lambda: ops.gemm(a, w, bias=b32)):6.1f}"
        print(line, flush=True)


def store_wt():
    """Write-through (sc1) output stores against plain ones, per shape of the step, as chains of n dependent-by-stream launches replayed
    from one hipGraph (every launch's boundary -- the write-back of the lines its predecessor left dirty -- is inside the time)."""
    for name, M, N, K, epi in SHAPES:
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        b32 = rnd(N, seed=3)
        r = rnd(M, N, seed=4).to(torch.bfloat16) if epi == "res" else None
        st, ws = ops.row_stats(a), w.float().sum(1).contiguous()
        if epi == "res":
            fn = lambda: ops.gemm(a, w, bias=b32, res=r, want_stats=True)
        elif epi == "geglu":
            fn = lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5), geglu=True)
        elif epi == "ln":
            fn = lambda: ops.gemm(a, w, bias=b32, ln=(st, ws, 1e-5))
        else:
            fn = lambda: ops.gemm(a, w, bias=b32)
        ts = {0: [], 1: []}
        outs = {}
        for rep in range(3):
            for wt in (0, 1):
                ENV["CD360_STORE_WT"] = str(wt)
                ts[wt].append(timeit_graph(fn, n=20))
                o = fn()
                outs[wt] = (o[0] if isinstance(o, tuple) else o).clone()
        same = torch.equal(outs[0], outs[1])
        print(f"{name:8s} M={M:6d} N={N:5d} K={K:4d} {epi:5s} | plain {min(ts[0]):7.1f} us | write-through {min(ts[1]):7.1f} us "
              f"({(min(ts[1]) / min(ts[0]) - 1) * 100:+.1f} %) | outputs bit-identical: {same}", flush=True)
    ENV.pop("CD360_STORE_WT", None)


def intercept():
    """Per-launch fixed cost of the one-round shapes: time against K (1 .. 80 K-tiles) at M = 3072, N = 1280 for the epilogue variants
    the step uses; a straight-line fit gives the time per K-tile and the intercept (launch boundary + prologue + epilogue)."""
    M, N = 3072, 1280
    for label, kw in (("plain", {}), ("bias", {"bias": True}), ("bias+res+stats", {"bias": True, "res": True, "stats": True}),
                      ("ln+bias", {"bias": True, "ln": True})):
        pts = []
        for K in (64, 128, 256, 640, 1280, 2560, 5120):
            a = rnd(M, K, seed=1).to(torch.bfloat16)
            w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
            b32 = rnd(N, seed=3) if kw.get("bias") else None
            r = rnd(M, N, seed=4).to(torch.bfloat16) if kw.get("res") else None
            ln = (ops.row_stats(a), w.float().sum(1).contiguous(), 1e-5) if kw.get("ln") else None
            fn = lambda: ops.gemm(a, w, bias=b32, res=r, ln=ln, want_stats=bool(kw.get("stats")))
            pts.append((K // 64, timeit_graph(fn, n=30)))
        n = len(pts)
        sx, sy = sum(p[0] for p in pts[2:]), sum(p[1] for p in pts[2:])
        sxx, sxy = sum(p[0] ** 2 for p in pts[2:]), sum(p[0] * p[1] for p in pts[2:])
        m = n - 2
        slope = (m * sxy - sx * sy) / (m * sxx - sx * sx)
        icpt = (sy - slope * sx) / m
        print(f"{label:15s}: " + "  ".join(f"K={64 * k:4d} {t:6.1f}" for k, t in pts) + f"  | fit over K >= 256: {slope:.3f} us per K-tile + {icpt:.1f} us", flush=True)
    # the floor: a kernel with nothing to do, chained the same way
    x = torch.zeros(64, device="cuda")
    print(f"empty-ish torch kernel chain (x.add_(1) on 64 floats): {timeit_graph(lambda: x.add_(1.0), n=50):.2f} us per launch", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    good = True
    if "intercept" in what:
        intercept()
    if "store_wt" in what:
        store_wt()
    if "movers" in what:
        good = movers() and good
    if "ff1" in what:
        ff1_tilings()
    if "movers_all" in what:
        good = movers_all() and good
    if "stride" in what:
        stride_sweep()
    if "whatif" in what:
        whatif()
    if "mid_m" in what:
        mid_m()
    if "geglu_cost" in what:
        geglu_cost()
    if "small_m" in what:
        small_m()
    if "fixed" in what:
        fixed_cost()
    if "narrow" in what:
        narrow()
    if "ksplit" in what:
        good = ksplit()
    if "check" in what:
        good = check()
    if "qcheck" in what:
        good = check_qattn_all() and good
    if "qcheck8" in what:
        good = check_qattn_fp8_all() and good
    if "time" in what:
        time_all()
    if "qattn" in what:
        time_qattn()
    if "ablate" in what:
        ablate()
    if "ablate_small" in what:
        ablate_small()
    if "group_m" in what:
        group_m_sweep()
    if "qattn_one" in what:  # a few launches of the fused pose-token attention (A3) at both pose levels for rocprofv3 --pmc passes
        for b, nq, C in ((3, 98304, 640), (3, 24576, 1280), (3, 4096, 640), (3, 1024, 1280)):  # A3 at both pose levels, then A2 at both levels
            a = rnd(b, nq, C, seed=1).to(torch.bfloat16)
            w = rnd(C, C, seed=2, scale=C ** -0.5).to(torch.bfloat16)
            kv = rnd(b, 80, 2 * C, seed=4).to(torch.bfloat16)
            st, ws, cb = ops.row_stats(a), w.float().sum(1).contiguous(), rnd(C, seed=3)
            for _ in range(4):
                ops.qproj_attention(a, w, kv[..., :C], kv[..., C:], 77, C // 64, bias=cb, ln=(st, ws, 1e-5))
        torch.cuda.synchronize()
    if "one" in what:  # a few launches of one shape for rocprofv3 --pmc passes: one M N K (env CD360_GEMM_* select the variant)
        M, N, K = (int(v) for v in what[what.index("one") + 1:what.index("one") + 4])
        a = rnd(M, K, seed=1).to(torch.bfloat16)
        w = rnd(N, K, seed=2, scale=K ** -0.5).to(torch.bfloat16)
        for _ in range(6):
            ops.gemm(a, w)
        torch.cuda.synchronize()
    sys.exit(0 if good else 1)
