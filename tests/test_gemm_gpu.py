"""cd360_gemm_bf16 / cd360_qproj_attn*_bf16 through the C ABI against fp32 torch references (`-m gpu`): every epilogue (bias, residual,
LayerNorm fold, GEGLU, row statistics), every tiling, ragged shapes, bit-identical repeat launches; the fused q-projection + attention
kernel incl. its de-duplicated CFG form.  The case lists live in tools/bench_gemm.py (the same functions time the kernels)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "custom-diffusion360_amd"))

pytestmark = pytest.mark.gpu


def test_gemm_family_and_fused_qproj_attention_against_fp32_torch(tune):
    import bench_gemm
    assert bench_gemm.check()


def test_fused_qproj_attention_fp8_mfma_variant_against_fp32_torch(tune):
    """BASELINE configs[4] inside the product kernel: cd360_kv_pack_fp8 + cd360_qproj_attn_fp8_bf16 (q K^T and P V on
    v_mfma_scale_f32_32x32x64_f8f6f4) on every tile, with and without the de-duplicated CFG form, against fp32 torch (layout / scale guard at 1.2e-1 of the
    output's max magnitude: e4m3 logits measure 6e-2 ... 1e-1 on unit-scale inputs, the bf16 kernel 4e-3 -- printed side by side), and the
    de-duplicated CFG form bit-identical to the expanded batch."""
    import bench_gemm
    assert bench_gemm.check_qattn_fp8_all()


def test_gemm_k_step_groups_of_the_128x128_tiling(tune):
    """The in-workgroup split of K (two groups of waves on alternate k-step pairs, partial sums exchanged through the LDS): parity of
    every epilogue it serves against fp32 torch, the convolution form against the unsplit kernel, repeat-equal launches."""
    import bench_gemm
    assert bench_gemm.ksplit(time=False)


def test_gemm_mover_waves_change_no_bit(tune):
    """cd360_tuning.gemm_movers = 4 (four extra waves issue every LDS-DMA piece, the others only multiply) against 0 on the four-buffer 128 x 128
    tiling, unsplit and with the k-step groups, Linear and convolution epilogues, ragged shapes: outputs and statistics bit-identical;
    and the default dispatch (movers on where they help) passes the family's parity list in the test above."""
    import bench_gemm
    assert bench_gemm.movers(time=False)


@pytest.mark.parametrize("M,N,K,geglu,ln,res", [(256, 256, 64, False, False, False), (256, 256, 128, False, False, True), (300, 272, 192, False, True, True),
                                                  (512, 768, 320, True, True, False), (1024, 512, 1280, True, False, False), (3072, 10240, 1280, True, True, False),
                                                  (2000, 1280, 640, False, True, True), (4096, 4096, 4096, False, False, False), (640, 2560, 448, False, False, False)])
def test_four_wave_generated_loop_changes_no_bit(tune, M, N, K, geglu, ln, res):
    """gemm_cfg = 9 -- 256 x 256 tiles as four waves of 128 x 128 on the generated instruction stream of csrc/gemm4w_loop.inc (operands over a
    five-slot LDS ring, tools/gen_gemm4w_loop.py) -- against the eight-wave arrangement of the same tile (gemm_cfg = 3): the same products
    summed in the same order, so the outputs are bit-identical (the row statistics to fp32 rounding); one to 64 K-tiles (every residue of the ring's five-tile
    period, loops that leave after the first trip), ragged M and N, every epilogue it serves (bias, LayerNorm fold, residual, GEGLU, row
    statistics); repeated launches equal; and against fp32 torch."""
    from bench_gemm import rnd
    from cd360 import ops
    a = rnd(M, K, seed=41).to(torch.bfloat16)
    w = rnd(N, K, seed=42, scale=K ** -0.5)
    b = rnd(N, seed=43)
    r = rnd(M, N, seed=44).to(torch.bfloat16) if res else None
    if ln:
        a = (a.float() * (0.5 + rnd(M, 1, seed=45).abs()) + 0.7 * rnd(M, 1, seed=46)).to(torch.bfloat16)
        gamma, beta = 1 + 0.2 * rnd(K, seed=47), 0.1 * rnd(K, seed=48)
        wk, wsum, cb = ops.pack_ln_linear(w, b, gamma, beta)
        want = torch.nn.functional.linear(torch.nn.functional.layer_norm(a.float(), (K,), gamma, beta, 1e-5), w, b)
        kw = dict(bias=cb, ln=(ops.row_stats(a), wsum, 1e-5))
    else:
        wk = w.to(torch.bfloat16)
        want = torch.nn.functional.linear(a.float(), wk.float(), b)
        kw = dict(bias=b)
    if geglu:
        perm = ops.geglu_row_order(N // 2, a.device)
        wk = wk[perm].contiguous()
        kw = {k: ((v[0], v[1][perm].contiguous(), v[2]) if k == "ln" else v[perm].contiguous()) for k, v in kw.items()}
        kw["geglu"] = True
        want = want[:, :N // 2] * torch.nn.functional.gelu(want[:, N // 2:])
    else:
        kw["want_stats"] = True
        if res:
            kw["res"] = r
            want = want + r.float()

    def run(cfg):
        tune(gemm_cfg=cfg)
        o = ops.gemm(a, wk, **kw)
        torch.cuda.synchronize()
        return o if isinstance(o, tuple) else (o,)
    ref = run(3)
    outs = [run(9) for _ in range(3)]
    def same(o, q):  # outputs bit for bit; the row statistics are sums over 2 instead of 4 waves' partials: equal to fp32 rounding
        return torch.equal(o[0], q[0]) and all(((x - y).abs().max() <= 1e-5 * y.abs().max()).item() for x, y in zip(o[1:], q[1:]))
    for o in outs:
        assert same(o, ref) and all(torch.equal(x, y) for x, y in zip(o, outs[0]))
    assert ((outs[0][0].float() - want).abs().max() / want.abs().max()).item() < 8e-3
    tune(gemm_cfg=-1, gemm_asm4=1)  # the switch the dispatch reads: every launch that chose a 256 x 256 tiling moves onto the generated loop
    o = ops.gemm(a, wk, **kw)
    if ops._lib.load().cd360_gemm_tile_n(M, N) == 256:
        assert same(o if isinstance(o, tuple) else (o,), ref)


@pytest.mark.parametrize("M,N,K,cfg,geglu", [(3072, 1280, 1280, -1, False), (3072, 3840, 1280, -1, False), (3072, 10240, 1280, -1, True), (1024, 512, 256, 9, False),
                                               (1024, 1280, 1280, -1, False), (256, 320, 128, -1, False)])
def test_bias_and_wsum_need_only_the_abi_alignment(tune, M, N, K, cfg, geglu):
    """The epilogue fetches the tile's slices of bias and wsum by LDS-DMA (16 bytes per lane, round 6).  The C ABI promises 8-byte alignment
    for both (cd360_hip.h): a slice at element offset 2 of a larger fp32 tensor must give the result of an aligned copy, bit for bit, on
    every tiling family (128 x 128 with movers, 256 x 192, sixteen-wave GEGLU, the generated four-wave loop, 64 x 128, a ragged 320-wide one)."""
    from bench_gemm import rnd
    from cd360 import ops
    a = rnd(M, K, seed=51).to(torch.bfloat16)
    w = rnd(N, K, seed=52, scale=K ** -0.5)
    gamma, beta = 1 + 0.2 * rnd(K, seed=53), 0.1 * rnd(K, seed=54)
    wp, wsum, cb = ops.pack_ln_linear(w, rnd(N, seed=55), gamma, beta)
    kw = {}
    if geglu:
        perm = ops.geglu_row_order(N // 2, a.device)
        wp, wsum, cb = wp[perm].contiguous(), wsum[perm].contiguous(), cb[perm].contiguous()
        kw["geglu"] = True
    bigw, bigc = torch.zeros(N + 8, device=a.device), torch.zeros(N + 8, device=a.device)
    bigw[2:2 + N], bigc[2:2 + N] = wsum, cb
    assert bigw[2:].data_ptr() % 16 == 8 and bigc[2:].data_ptr() % 16 == 8
    tune(gemm_cfg=cfg)
    st = ops.row_stats(a)
    want = ops.gemm(a, wp, bias=cb, ln=(st, wsum, 1e-5), **kw)
    got = ops.gemm(a, wp, bias=bigc[2:2 + N], ln=(st, bigw[2:2 + N], 1e-5), **kw)
    assert torch.equal(want, got)


@pytest.mark.parametrize("ksplit", [0, 1])
def test_gemm_cstats_in_both_wave_arrangements(tune, ksplit):
    from bench_gemm import rnd
    from cd360 import ops
    tune(gemm_ksplit=ksplit)
    M, N, K = 3072, 1280, 1280
    a, w = rnd(M, K, seed=31).to(torch.bfloat16), rnd(N, K, seed=32, scale=K ** -0.5).to(torch.bfloat16)
    b, r = rnd(N, seed=33), rnd(M, N, seed=34).to(torch.bfloat16)
    out, cst = ops.gemm_cstats(a, w, bias=b, res=r)
    assert cst is not None and torch.equal(out, ops.gemm(a, w, bias=b, res=r))
    ref = out.float().reshape(M // 64, 64, N)
    rel = lambda x, y: ((x - y).abs().max() / y.abs().max().clamp_min(1e-6)).item()
    assert rel(cst[..., 0], ref.sum(1)) < 1e-5 and rel(cst[..., 1], (ref * ref).sum(1)) < 1e-5


@pytest.mark.parametrize("b,dup,nq,C,nk", [(2, 1, 256, 128, 77), (4, 2, 512, 640, 77), (2, 1, 1024, 1280, 50), (1, 1, 256, 64, 20)])
def test_qproj_attention_dedup_equals_the_expanded_batch(b, dup, nq, C, nk):
    """cd360_qproj_attn_dedup_bf16: the last `dup` query batch elements meet two key / value sets (batch i and i + dup).  Must equal,
    bit for bit, cd360_qproj_attn_bf16 on the batch with those elements repeated (sample.py's 3-way CFG: [null | image | image+text])."""
    from bench_gemm import rnd
    from cd360 import ops
    heads, K = C // 64, C
    a = (rnd(b, nq, K, seed=21) * (0.5 + rnd(b, nq, 1, seed=22).abs()) + 0.5 * rnd(b, nq, 1, seed=23)).to(torch.bfloat16)
    gamma, beta = 1 + 0.2 * rnd(K, seed=5), 0.1 * rnd(K, seed=6)
    wp, wsum, cb = ops.pack_ln_linear(rnd(C, K, seed=24, scale=K ** -0.5), None, gamma, beta)
    kv = rnd(b + dup, max(80, nk), 2 * C, seed=25).to(torch.bfloat16)
    k, v = kv[..., :C], kv[..., C:]
    got = ops.qproj_attention(a, wp, k, v, nk, heads, bias=cb, ln=(ops.row_stats(a), wsum, 1e-5), dup=dup)
    a3 = torch.cat([a, a[b - dup:]], 0).contiguous()
    want = ops.qproj_attention(a3, wp, k, v, nk, heads, bias=cb, ln=(ops.row_stats(a3), wsum, 1e-5))
    assert got.shape == (b + dup, nq, C) and torch.equal(got, want)


@pytest.mark.parametrize("C,nq", [(640, 98304), (1280, 24576)])
def test_qproj_attention_dedup_at_the_headline_size_matches_fp32_on_row_slices(C, nq):
    """cd360_qproj_attn_dedup_bf16 at the pose-token count of BASELINE configs[1] (1024^2 image: 4096 x 24 = 98 304 tokens per CFG branch at
    the 640 level, 1024 x 24 at the 1280 level; b = 2 + 1 de-duplicated, 77 text keys, LayerNorm fold): 4096-row slices from the start,
    the middle and the end of every output batch element against fp32 torch (attention.py:578-588 = LayerNorm -> to_q -> softmax(q k^T / 8) v)."""
    import torch.nn.functional as F
    from bench_gemm import rnd
    from cd360 import ops
    b, dup, nk, heads, K = 2, 1, 77, C // 64, C
    a = (rnd(b, nq, K, seed=41) * (0.5 + rnd(b, nq, 1, seed=42).abs()) + 0.5 * rnd(b, nq, 1, seed=43)).to(torch.bfloat16)
    gamma, beta = 1 + 0.2 * rnd(K, seed=5), 0.1 * rnd(K, seed=6)
    w = rnd(C, K, seed=44, scale=K ** -0.5)
    wp, wsum, cb = ops.pack_ln_linear(w, None, gamma, beta)
    kv = rnd(b + dup, 80, 2 * C, seed=45).to(torch.bfloat16)
    k, v = kv[..., :C], kv[..., C:]
    got = ops.qproj_attention(a, wp, k, v, nk, heads, bias=cb, ln=(ops.row_stats(a), wsum, 1e-5), dup=dup)
    assert got.shape == (b + dup, nq, C) and bool(torch.isfinite(got).all())
    worst = 0.0
    for ob in range(b + dup):
        qb = ob if ob < b else ob - dup  # query element of output batch ob
        for r0 in (0, nq // 2 - 2048, nq - 4096):
            x = a[qb, r0:r0 + 4096].float()
            q = F.linear(F.layer_norm(x, (K,), gamma, beta, 1e-5), w).reshape(4096, heads, 64).transpose(0, 1)
            kh = k[ob, :nk].float().reshape(nk, heads, 64).transpose(0, 1)
            vh = v[ob, :nk].float().reshape(nk, heads, 64).transpose(0, 1)
            want = (torch.softmax(q @ kh.transpose(-1, -2) / 8.0, -1) @ vh).transpose(0, 1).reshape(4096, C)
            worst = max(worst, ((got[ob, r0:r0 + 4096].float() - want).abs().max() / want.abs().max()).item())
    print(f"qproj_attn_dedup C={C} nq={nq}: worst slice error {worst:.2e}")
    assert worst < 1e-2


@pytest.mark.parametrize("M,N,K", [(3072, 1280, 1280), (12288, 640, 640), (192, 64, 128), (1024, 1280, 1280), (1024, 1280, 5120)])
def test_gemm_with_groupnorm_channel_statistics(M, N, K):
    """cd360_gemm_cstats_bf16 (SpatialTransformer.proj_out + residual feeding a GroupNorm): same output as cd360_gemm_bf16, per-slab
    (64 rows; 32 on the 64 x 128 tiling of small batches, both wave arrangements) channel sums / sums of squares of the stored values
    exact, deterministic."""
    from bench_gemm import rnd
    from cd360 import ops
    a = rnd(M, K, seed=31).to(torch.bfloat16)
    w = rnd(N, K, seed=32, scale=K ** -0.5).to(torch.bfloat16)
    b = rnd(N, seed=33)
    r = rnd(M, N, seed=34).to(torch.bfloat16)
    out, cst = ops.gemm_cstats(a, w, bias=b, res=r)
    assert torch.equal(out, ops.gemm(a, w, bias=b, res=r))
    if cst is None:
        pytest.skip("tiling without row slabs for this shape")
    slab = M // cst.shape[0]
    assert slab in (32, 64) and cst.shape == (M // slab, N, 2)
    ref = out.float().reshape(M // slab, slab, N)
    rel = lambda x, y: ((x - y).abs().max() / y.abs().max().clamp_min(1e-6)).item()
    assert rel(cst[..., 0], ref.sum(1)) < 1e-5 and rel(cst[..., 1], (ref * ref).sum(1)) < 1e-5
    assert torch.equal(cst, ops.gemm_cstats(a, w, bias=b, res=r)[1])
