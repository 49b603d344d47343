"""Parity of every HIP kernel (called through the C ABI via cd360.ops) against the CPU oracle.  Needs an MI355X.

Tolerances: integer indices / fp32 coordinate chains bit-exact; bf16 kernels within 1e-2 of the oracle relative to the
tensor's max magnitude (BASELINE.json north_star), on inputs rounded to bf16 for both sides so only kernel error is measured."""
import math

import numpy as np
import pytest
import torch

import weights as W
from oracle import pose_path as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.isfinite(got).all()
    return (got - want).abs().max().item() / max(want.abs().max().item(), 1e-12)


def bf(x):
    return x.to(torch.bfloat16).float()


def cams_for(b, n, seed):
    from cd360 import synth
    from cd360.cameras import pack_cameras
    return pack_cameras(synth.pose_batch(b, n, seed=seed))


# ------------------------------------------------------------------------------------------------ attention (A1-A3)
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 2, 128, 128), (1, 3, 200, 77), (1, 1, 1024, 1024), (2, 2, 96, 40), (1, 10, 6144, 77), (1, 2, 333, 333),
                                        (1, 2, 50, 20), (1, 1, 33, 96), (1, 2, 100, 97), (3, 2, 4100, 77)])
def test_attention_strided(B, H, Nq, Nk):
    from cd360 import ops
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q = bf(torch.randn(B, Nq, H * 64, generator=g))
    k = bf(torch.randn(B, Nk, H * 64, generator=g))
    v = bf(torch.randn(B, Nk, H * 64, generator=g))
    nkp = (Nk + 7) // 8 * 8
    # k and v as column slices of one wider row-major tensor (the merged projection layout), NaN rows beyond Nk: padding and the
    # neighbouring slice must never be read as data
    kv = torch.full((B, nkp, 2 * H * 64 + 64), float("nan"))
    kv[:, :Nk, :H * 64] = k
    kv[:, :Nk, H * 64 + 64:] = v
    kvd = kv.to(DEV, torch.bfloat16)
    out = ops.attention(q.to(DEV, torch.bfloat16), kvd[..., :H * 64], kvd[..., H * 64 + 64:], H, nk=Nk)

    def split(t):
        return t.reshape(B, t.shape[1], H, 64).permute(0, 2, 1, 3).reshape(B * H, t.shape[1], 64)

    want = O.attention_core(split(q), split(k), split(v)).reshape(B, H, Nq, 64).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    assert rel(out, want) < 1e-2


def test_attention_xformers_layout_and_online_softmax_rescale():
    """xformers-layout entry point; a spiked key late in the sequence forces the running-max rescale branch."""
    from cd360 import ops
    g = torch.Generator().manual_seed(5)
    q, k, v = (bf(torch.randn(4, 300, 64, generator=g)) for _ in range(3))
    k[:, 257] = 6.0 * q[:, 10]  # key 257 (tile 4) dominates query 10
    out = ops.memory_efficient_attention(q.to(DEV, torch.bfloat16), k.to(DEV, torch.bfloat16), v.to(DEV, torch.bfloat16))
    assert rel(out, O.attention_core(q, k, v)) < 1e-2


@pytest.mark.parametrize("neg", [3.0, 16.0])
@pytest.mark.parametrize("gen", ["0", "1", "2"])
@pytest.mark.parametrize("B,H,N", [(1, 2, 256), (2, 3, 1024), (1, 1, 4096)])
def test_self_attention_generations_agree_with_oracle(gen, B, H, N, neg, tune):
    """The whole-tile self-attention kernel (attn_self_kernel: lazy running maximum, pre-scaled q, LDS-DMA ring; CD360_ATTN_SELF =
    1: 4 waves x 32 queries, 2: 8 waves x 64 queries) and the first-generation tiled kernel (0) against the fp32 oracle, on the merged
    q|k|v layout.
    Query 10 meets a dominating key late in the sequence (m_ref must move mid-stream), query 20 starts on a tile of strongly
    negative scores (m_ref must move UP by far more than the 2^8 slack right after the first tile; with neg = 16 every score of its
    first tile is below -128 in exp2 units, where an unclamped first rescale factor exp2(-max) overflows to inf and 0 * inf poisons the
    row); lse as the training forward returns it."""
    from cd360 import ops
    if gen == "2" and N % 512:
        pytest.skip("the 8-wave x 64-query variant needs 512 queries per workgroup")
    tune(attn_self=int(gen))
    g = torch.Generator().manual_seed(N + H)
    qkv = bf(torch.randn(B, N, 3 * H * 64, generator=g))
    q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
    k[:, N - 70] = 8.0 * q[:, 10]
    k[:, :64] = -neg * q[:, 20:21]
    qkv = bf(qkv)
    d = qkv.to(DEV, torch.bfloat16)
    out, lse = ops.attention(d[..., :H * 64], d[..., H * 64:2 * H * 64], d[..., 2 * H * 64:], H, want_lse=True)
    assert torch.isfinite(out.float()).all() and torch.isfinite(lse).all()

    def split(t):
        return t.reshape(B, N, H, 64).permute(0, 2, 1, 3).reshape(B * H, N, 64)

    qs, ks, vs = split(qkv[..., :H * 64]), split(qkv[..., H * 64:2 * H * 64]), split(qkv[..., 2 * H * 64:])
    want = O.attention_core(qs, ks, vs).reshape(B, H, N, 64).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    assert rel(out, want) < (1e-2 if gen == "0" else 1.5e-2)  # in-kernel q pre-scaling costs one more bf16 rounding (the modules fold it into Wq)
    want_lse = torch.logsumexp(torch.einsum("bqd,bkd->bqk", qs.double(), ks.double()) / 8.0, dim=-1).float()
    assert (lse.cpu() - want_lse).abs().max().item() < 2e-2 * max(1.0, want_lse.abs().max().item() / 10)


@pytest.mark.parametrize("B,H,N", [(2, 3, 1024), (1, 2, 4096), (1, 2, 200), (3, 2, 512)])
def test_self_attention_prescaled_q_is_exact_to_the_bf16_bar(B, H, N):
    """cd360_attn_fwd_prescaled_bf16 (what the transformer blocks call: the softmax scale and log2 e live in the q projection):
    softmax_2(q' k^T) v against the oracle's softmax(q k^T / 8) v with q = q' / (log2 e / 8) -- no extra rounding, the plain 1e-2 bar;
    whole-tile shapes take attn_self_kernel (both tilings), the ragged one the first-generation kernel."""
    from cd360 import ops
    g = torch.Generator().manual_seed(7 * N + H)
    qkv = bf(torch.randn(B, N, 3 * H * 64, generator=g))
    qkv[..., :H * 64] *= 0.5  # q' = q * 0.18: keep the logits in the usual range
    qkv[:, N - 70, H * 64:2 * H * 64] = 30.0 * qkv[:, 10, :H * 64]
    qkv = bf(qkv)
    d = qkv.to(DEV, torch.bfloat16)
    out = ops.attention(d[..., :H * 64], d[..., H * 64:2 * H * 64], d[..., 2 * H * 64:], H, prescaled=True)

    def split(t):
        return t.reshape(B, N, H, 64).permute(0, 2, 1, 3).reshape(B * H, N, 64)

    want = O.attention_core(split(qkv[..., :H * 64]) / ops.ATTN_PRESCALE, split(qkv[..., H * 64:2 * H * 64]), split(qkv[..., 2 * H * 64:]))
    assert rel(out, want.reshape(B, H, N, 64).permute(0, 2, 1, 3).reshape(B, N, H * 64)) < 1e-2


# ------------------------------------------------------------------------------------------------ rays / indices (A4, A5)
@pytest.mark.parametrize("b,n,r,S,jitter", [(2, 3, 8, 4, False), (1, 4, 32, 24, False), (2, 2, 16, 24, True), (1, 8, 64, 24, False)])
def test_rays_points_grid_and_indices_bit_exact(b, n, r, S, jitter):
    from cd360 import nerf, ops
    cams = cams_for(b, n, seed=r + S)
    jx = W.uniform("jx", (r + 1,), seed=r) if jitter else None
    jy = W.uniform("jy", (r + 1,), seed=r) if jitter else None
    jd = W.uniform("jd", (r * r, S + 1), seed=r) if jitter else None
    xs_o, ys_o = O.patch_positions(r, jx), O.patch_positions(r, jy)
    rays_o = O.patch_rays(cams, xs_o, ys_o)
    len_o, _ = O.depth_samples(S, 2.0, 0.0, jd, r * r)
    pts_o = O.ray_points(rays_o, len_o)
    grid_o = O.sample_grid(cams, pts_o)
    x0_o, y0_o, _, _, m_o = O.bilinear_corners(grid_o, r)

    xs, ys = nerf.patch_positions(r, DEV, jx), nerf.patch_positions(r, DEV, jy)
    t, _ = nerf.depth_samples(S, 2.0, 0.0, DEV, r * r, jd)
    cd = cams.to(DEV)
    assert torch.equal(ops.patch_rays(cd, xs, ys).cpu(), rays_o)
    res = ops.ray_project_index(cd, xs, ys, t)
    assert torch.equal(res["points"].cpu(), pts_o)
    assert torch.equal(res["grid"].cpu(), grid_o)
    assert torch.equal(res["x0"].cpu(), x0_o) and torch.equal(res["y0"].cpu(), y0_o) and torch.equal(res["mask"].cpu(), m_o)
    assert int((m_o != 15).sum()) > 0 or r <= 8  # the case set does exercise out-of-bounds corners


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_feature_gather(dtype):
    from cd360 import ops
    g = torch.Generator().manual_seed(3)
    n_img, r, C, P = 5, 16, 96, 700
    xref = torch.randn(n_img, r * r, C, generator=g)
    grid = (torch.rand(n_img, P, 2, generator=g) * 2.6 - 1.3).clamp(-1.2, 1.2)
    grid[0, :4] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [1.2, 0.0], [0.0, -1.2]])
    if dtype == torch.bfloat16:
        xref = bf(xref)
    want = O.gather_bilinear(xref[:, None], grid[:, None, :, None, :])[:, 0, :, 0]
    got = ops.feature_gather(xref.to(DEV, dtype), grid.to(DEV))
    assert rel(got, want) < (1e-6 if dtype == torch.float32 else 8e-3)


# ------------------------------------------------------------------------------------------------ fused FeatureNeRF (A5-A9)
def nerf_weights(C, seed):
    shapes = {"model.plane_coefs.0.weight": (C, C + 198), "model.plane_coefs.0.bias": (C,), "model.plane_coefs.2.weight": (C, C),
              "model.plane_coefs.2.bias": (C,), "model.nviews.weight": (1, C + 198), "model.nviews.bias": (1,), "model.decoder.weight": (4, C)}
    return {k[len("model."):]: v for k, v in W.synth_state_dict(shapes, seed).items()}


def test_plucker_features():
    from cd360 import nerf, ops
    b, n, r = 2, 3, 8
    cams = cams_for(b, n, seed=21)
    xs = O.patch_positions(r)
    rays = O.patch_rays(cams, xs, xs)
    tgt = rays[:, 0]
    cam_o = O.world_to_view(cams[:, 1:, None, :], tgt[:, None, :, :3])
    cam_d = O.rotate_to_view(cams[:, 1:, None, :], tgt[:, None, :, 3:])
    want = torch.cat([O.positional_encoding(O.plucker(torch.cat([cam_o, cam_d], -1)), 8), cam_d], -1)
    got = ops.plucker_features(cams.to(DEV), nerf.patch_positions(r, DEV), nerf.patch_positions(r, DEV)).cpu()
    assert torch.all(got[..., 99:] == 0)
    assert (got[..., :99] - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("C,r,n,S,b", [(64, 8, 2, 4, 2), (128, 16, 5, 24, 1), (640, 8, 3, 6, 1)])
def test_fused_feature_nerf(C, r, n, S, b):
    from cd360 import nerf
    w = nerf_weights(C, seed=C + n)
    cams = cams_for(b, n, seed=C)
    xref = bf(W.tensor("xref", (b, n, r * r, C), seed=C))
    feats, sigma, _, attn, rgb, _ = O.nerf_module(w, cams, xref, S, 2.0)
    fw = nerf.FusedNerfWeights(*(w[k].to(DEV) for k in ("plane_coefs.0.weight", "plane_coefs.0.bias", "plane_coefs.2.weight", "plane_coefs.2.bias",
                                                        "nviews.weight", "nviews.bias", "decoder.weight")))
    h, dec, dists, vw = nerf.fused_feature_nerf(fw, cams.to(DEV), xref.to(DEV, torch.bfloat16), S, 2.0, want_view_weights=True)
    assert rel(vw, attn) < 1e-2
    assert rel(h, feats) < 1e-2
    assert rel(dec[..., 3:], sigma) < 1e-2 and rel(dec[..., :3], rgb) < 1e-2


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("C,r,n,S,b", [(64, 8, 2, 4, 2), (640, 16, 5, 24, 1), (1280, 8, 7, 6, 2), (128, 7, 3, 3, 1)])
def test_render_kernel_full_line_gathers_equal_register_gathers(C, r, n, S, b, variant, tune):
    """The render kernel's full-line gather variants (cd360_tuning.nerf_kernel) against nerf_fused_kernel (per-lane register gathers):
    the same additions in the same order -- bit-identical g, logits and lse on eight repeated launches.  Variant 1: corner rows as
    full 128-byte lines through the wave's own LDS block (ordinary loads + ds_write / ds_read, ordered by the wave's own counters).
    Variant 2 (probe builds with -DCD360_WHATIF only): rows by LDS-DMA with one vmcnt(0) + workgroup barrier per view -- without
    that barrier a few rows per thousand launches came back stale, and this test is the detector.  (128, 7, ...) has a ragged last
    sample tile."""
    from cd360 import _lib
    if variant == 2 and not _lib.load().cd360_whatif_build():
        pytest.skip("the LDS-DMA render kernel exists in -DCD360_WHATIF probe builds only")
    from cd360 import nerf, ops
    w = nerf_weights(C, seed=C + n)
    cams = cams_for(b, n, seed=C).to(DEV)
    xref = bf(W.tensor("xref", (b, n, r * r, C), seed=C)).to(DEV, torch.bfloat16)
    fw = nerf.FusedNerfWeights(*(w[k].to(DEV) for k in ("plane_coefs.0.weight", "plane_coefs.0.bias", "plane_coefs.2.weight", "plane_coefs.2.bias",
                                                        "nviews.weight", "nviews.bias", "decoder.weight")))
    xs = nerf.patch_positions(r, DEV)
    t, _ = nerf.depth_samples(S, 2.0, 0.0, DEV, r * r)
    Y, lv = nerf.reference_tables(fw, xref)
    g = torch.Generator().manual_seed(C)
    zP = bf(torch.randn(b * n, r * r, C, generator=g)).to(DEV, torch.bfloat16)
    cview = nerf.view_constants(fw, cams)
    tune(nerf_kernel=0)
    ref = ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk, want_logits=True)
    tune(nerf_kernel=variant)
    outs = [ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk, want_logits=True) for _ in range(8)]
    for o in outs:
        assert all(torch.equal(a, b_) for a, b_ in zip(o, ref))


@pytest.mark.parametrize("geometry", [3, 4])
@pytest.mark.parametrize("C,r,n,S,b", [(64, 8, 2, 4, 2), (640, 16, 5, 24, 1), (1280, 8, 7, 6, 2), (128, 7, 3, 3, 1), (64, 8, 1, 4, 1)])
def test_render_two_pass_equals_one_pass(C, r, n, S, b, geometry, tune):
    """The two-pass render (round 6: cd360_nerf_mlp_aggregate_ws = nerf_geom_kernel + nerf_fused_rec_kernel, the binding's default)
    against the one-pass full-line kernel (cd360_tuning.nerf_kernel = 1): the view logits are the same fp32 chain (bit-identical), the
    softmax statistics agree to fp32 rounding, and the aggregated features differ only through the softmax weights' rounding (exp2(lg - m)
    / l with the final maximum instead of the online rescales): a last-bit flip of a bf16 output here and there.  Eight repeated launches
    of the two-pass form are bit-identical (no atomics, no ordering freedom); (.., n = 1, ..) exercises the short record pipeline.
    geometry = cd360_tuning.nerf_kernel: 3 = pass 2 on 64 channels per workgroup, 4 = on 32 (the same arithmetic per channel: the two are
    bit-identical to each other)."""
    from cd360 import nerf, ops
    w = nerf_weights(C, seed=C + n)
    cams = cams_for(b, n, seed=C).to(DEV)
    xref = bf(W.tensor("xref", (b, n, r * r, C), seed=C)).to(DEV, torch.bfloat16)
    fw = nerf.FusedNerfWeights(*(w[k].to(DEV) for k in ("plane_coefs.0.weight", "plane_coefs.0.bias", "plane_coefs.2.weight", "plane_coefs.2.bias",
                                                        "nviews.weight", "nviews.bias", "decoder.weight")))
    xs = nerf.patch_positions(r, DEV)
    t, _ = nerf.depth_samples(S, 2.0, 0.0, DEV, r * r)
    Y, lv = nerf.reference_tables(fw, xref)
    g = torch.Generator().manual_seed(C)
    zP = bf(torch.randn(b * n, r * r, C, generator=g)).to(DEV, torch.bfloat16)
    cview = nerf.view_constants(fw, cams)
    tune(nerf_kernel=1)
    ref = ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk, want_logits=True)
    tune(nerf_kernel=7 - geometry)
    other = ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk, want_logits=True)
    tune(nerf_kernel=geometry)
    outs = [ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk, want_logits=True) for _ in range(8)]
    assert all(torch.equal(a, b_) for a, b_ in zip(other, outs[0]))           # the two pass-2 geometries agree bit for bit
    assert torch.equal(outs[0][1], ref[1])                                    # logits: bit-identical
    assert torch.allclose(outs[0][2], ref[2], rtol=2e-6, atol=2e-6)          # (max ln 2, sum)
    assert rel(outs[0][0], ref[0]) < 4e-3 and (outs[0][0] != ref[0]).float().mean() < 0.05  # a bf16 ulp, on few entries
    nolog = ops.nerf_mlp_aggregate(cams, xs, xs, t, Y, zP, lv, cview, fw.Wk)
    assert torch.equal(nolog[0], outs[0][0]) and nolog[1] is None
    for o in outs[1:]:
        assert all(torch.equal(a, b_) for a, b_ in zip(o, outs[0]))


# ------------------------------------------------------------------------------------------------ volume rendering (A10)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_volrender(dtype):
    from cd360 import ops
    g = torch.Generator().manual_seed(8)
    b, hw, S, C = 2, 50, 24, 64
    feats = torch.randn(b, hw, S, C, generator=g)
    if dtype == torch.bfloat16:
        feats = bf(feats)
    sigma_raw = torch.randn(b, hw, S, 1, generator=g) * 2
    sigma_raw[0, 0, 3] = 60.0  # overflowing density: exercises nan_to_num / saturation
    rgb_raw = torch.randn(b, hw, S, 3, generator=g)
    dists = torch.rand(hw, S, generator=g) * 0.1 + 0.04
    want = O.vol_render(feats, torch.exp(sigma_raw), dists[None, :, :, None], torch.sigmoid(rgb_raw))
    got = ops.volrender(feats.to(DEV, dtype), sigma_raw[..., 0].to(DEV), dists.to(DEV), rgb_raw.to(DEV), want_weights=True)
    tol = 1e-5 if dtype == torch.float32 else 5e-3
    for gi, wi in zip(got, want):
        assert rel(gi, wi) < tol
    # module-style call: densities already exponentiated, rgb already sigmoid'ed, shared [S] dists
    got2 = ops.volrender(feats.to(DEV, dtype), torch.exp(sigma_raw[..., 0]).to(DEV), dists[0].to(DEV), torch.sigmoid(rgb_raw).to(DEV),
                         sigma_is_raw=False, rgb_is_raw=False)
    want2 = O.vol_render(feats, torch.exp(sigma_raw), dists[0][None, None, :, None], torch.sigmoid(rgb_raw))
    assert rel(got2[0], want2[0]) < tol and rel(got2[4], want2[4]) < 1e-5


@pytest.mark.parametrize("C", [64, 520, 640, 1280, 2048])
def test_rowdot4(C):
    """decoder head (nerfsd_pytorch3d.py:49-51,160): every number of 512-channel chunks a lane can own, a ragged last chunk"""
    from cd360 import ops
    g = torch.Generator().manual_seed(9)
    h, w = bf(torch.randn(3, 37, C, generator=g)), torch.randn(4, C, generator=g)
    got = ops.rowdot4(h.to(DEV, torch.bfloat16), w.to(DEV))
    assert rel(got, h @ w.t()) < 1e-5 and torch.equal(got, ops.rowdot4(h.to(DEV, torch.bfloat16), w.to(DEV)))


@pytest.mark.parametrize("rows,C", [(111, 64), (1000, 640), (24576, 1280), (257, 520)])
def test_rowdot4_backward(rows, C):
    """cd360_rowdot4_bwd_bf16 (decoder head of the FeatureNeRF samples, nerfsd_pytorch3d.py decoder Linear): dh = d w per row, dw = d^T h
    reduced per 256-row slab (ragged last slab, channel counts that leave lanes of the last 512-channel group idle), deterministic."""
    from cd360 import ops
    g = torch.Generator().manual_seed(19)
    h, w, d = bf(torch.randn(rows, C, generator=g)), torch.randn(4, C, generator=g), torch.randn(rows, 4, generator=g)
    dh, dw = ops.rowdot4_bwd(d.to(DEV), h.to(DEV, torch.bfloat16), w.to(DEV))
    assert rel(dh, d @ w) < 1e-2 and rel(dw, d.t() @ h) < 1e-5
    assert torch.equal(dw, ops.rowdot4_bwd(d.to(DEV), h.to(DEV, torch.bfloat16), w.to(DEV), need_dh=False)[1])


def test_geglu_and_concat():
    from cd360 import ops
    g = torch.Generator().manual_seed(10)
    p = bf(torch.randn(3, 50, 2 * 320, generator=g) * 2)
    x, gate = p.chunk(2, dim=-1)
    assert rel(ops.geglu(p.to(DEV, torch.bfloat16)), x * torch.nn.functional.gelu(gate)) < 8e-3
    a, b = bf(torch.randn(2, 64, 5, 7, generator=g)), bf(torch.randn(2, 24, 5, 7, generator=g))
    cl = torch.channels_last
    got = ops.concat_channels(a.to(DEV, torch.bfloat16).contiguous(memory_format=cl), b.to(DEV, torch.bfloat16).contiguous(memory_format=cl))
    assert got.shape == (2, 88, 5, 7) and torch.equal(got.float().cpu(), torch.cat([a, b], 1))


def test_cfg_euler_step_kernel_matches_sampler_chain():
    from cd360.sampler import cfg_euler_update
    x, eps = W.tensor("x", (2, 4, 16, 16), seed=3), W.tensor("eps", (6, 4, 16, 16), seed=3)
    s, sn = torch.tensor([3.3]), torch.tensor([2.9])
    want = cfg_euler_update(x, eps, s, sn, 7.5, 3.5, fused=False)
    got = cfg_euler_update(x.to(DEV), eps.to(DEV), s.to(DEV), sn.to(DEV), 7.5, 3.5, fused=True)
    assert rel(got, want) < 1e-6


@pytest.mark.parametrize("rows,C", [(37, 64), (1000, 640), (513, 1280), (5, 2048)])
def test_add_layernorm(rows, C):
    from cd360 import ops
    g = torch.Generator().manual_seed(rows + C)
    a, b = bf(torch.randn(rows, C, generator=g) * 2), bf(torch.randn(rows, C, generator=g) + 0.3)
    gamma, beta = bf(torch.randn(C, generator=g)), bf(torch.randn(C, generator=g))
    s, ln = ops.add_layernorm(a.to(DEV, torch.bfloat16), b.to(DEV, torch.bfloat16), gamma.to(DEV, torch.bfloat16), beta.to(DEV, torch.bfloat16), 1e-5)
    assert rel(s, a + b) < 5e-3
    assert rel(ln, torch.nn.functional.layer_norm(a + b, (C,), gamma, beta, 1e-5)) < 8e-3
    s2, ln2 = ops.add_layernorm(a.to(DEV, torch.bfloat16), None, gamma.to(DEV, torch.bfloat16), beta.to(DEV, torch.bfloat16), 1e-5)
    assert s2 is None and rel(ln2, torch.nn.functional.layer_norm(a, (C,), gamma, beta, 1e-5)) < 8e-3


@pytest.mark.parametrize("N,H,W,Cin,Cout,taps,extras", [(2, 9, 7, 64, 48, 9, False), (1, 16, 16, 128, 320, 9, True), (3, 32, 32, 320, 640, 9, True),
                                                       (2, 5, 5, 192, 64, 1, True), (1, 1, 1, 64, 16, 9, False), (2, 9, 7, 128, 320, 9, True), (1, 12, 12, 64, 160, 9, False)])
@pytest.mark.parametrize("split", ["auto", "1", "2"])
def test_conv_igemm(N, H, W, Cin, Cout, taps, extras, split, tune):
    """implicit-GEMM conv3x3 / GEMM with fused bias + per-image addend + residual vs torch's fp32 conv2d on the same bf16 inputs.
    `split`: the in-workgroup split-K variant (512 threads, odd K-steps on the second 4 waves) forced on / off / chosen by the
    launch heuristic; (3,32,32,320,640) has an ODD number of K-steps (45), (1,16,16,128,320) the minimum of 2 chunks per tap.
    Cout = 320 / 160 take the 160-channel tiling (4 waves of 160 x 32), with ragged pixel tiles in (2,9,7,...)."""
    from cd360 import ops
    if split != "auto":  # the register-staged kernel and its split-K variant (3 x 3 / stride 1 otherwise runs on the LDS-DMA core)
        tune(conv_split=int(split), conv_dma=0)
    g = torch.Generator().manual_seed(N * 100 + H + Cin + Cout)
    k = 3 if taps == 9 else 1
    x = bf(torch.randn(N, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    emb = bf(torch.randn(N, Cout, generator=g)) if extras else None
    res = bf(torch.randn(N, Cout, H, W, generator=g)) if extras else None
    want = torch.nn.functional.conv2d(x, w, bias, padding=k // 2)
    if extras:
        want = want + emb[:, :, None, None] + res
    xt = x.permute(0, 2, 3, 1).reshape(N, H * W, Cin).contiguous().to(DEV, torch.bfloat16)
    wp = ops.pack_conv_weight(w).to(DEV)
    rt = None if res is None else res.permute(0, 2, 3, 1).reshape(N, H * W, Cout).contiguous().to(DEV, torch.bfloat16)
    got = ops.conv_igemm(xt, wp, bias.to(DEV), N, H, W, taps, None if emb is None else emb.to(DEV, torch.bfloat16), rt)
    assert rel(got.reshape(N, H, W, Cout).permute(0, 3, 1, 2), want) < 8e-3


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 128, 256), (1, 32, 32, 64, 320), (3, 32, 32, 320, 640), (2, 16, 8, 128, 160)])
def test_conv_epilogue_groupnorm_statistics(N, H, W, Cin, Cout):
    """conv_igemm(want_stats=True) hands the GroupNorm that follows its per-slab channel sums (openaimodel.py:352-376: conv ->
    GN -> SiLU -> conv): GN from those statistics must equal GN with its own statistics pass, in every launch shape (default
    128-channel tiles, 160-channel tiles for Cout = 320 / 160, in-workgroup split-K for (3,32,32,320,640)), with emb + residual."""
    from cd360 import ops
    g = torch.Generator().manual_seed(Cin + Cout)
    x = bf(torch.randn(N, H * W, Cin, generator=g)).to(DEV, torch.bfloat16)
    wp = ops.pack_conv_weight(bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    emb = bf(torch.randn(N, Cout, generator=g)).to(DEV, torch.bfloat16)
    res = bf(torch.randn(N, H * W, Cout, generator=g)).to(DEV, torch.bfloat16)
    out, stats = ops.conv_igemm(x, wp, bias, N, H, W, 9, emb, res, want_stats=True)
    assert torch.equal(out, ops.conv_igemm(x, wp, bias, N, H, W, 9, emb, res))  # same output with and without the statistics
    slabs = stats.shape[1]
    assert stats.shape == (N, slabs, Cout, 2) and (H * W) % slabs == 0
    want = out.float().reshape(N, slabs, H * W // slabs, Cout)
    assert rel(stats[..., 0], want.sum(2)) < 1e-5 and rel(stats[..., 1], (want * want).sum(2)) < 1e-5
    gamma, beta = torch.randn(Cout, generator=g).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    a = ops.gn_silu(out, gamma, beta, 32, 1e-5, True, tile_stats=stats)
    b = ops.gn_silu(out, gamma, beta, 32, 1e-5, True)
    assert rel(a, b) < 4e-3  # bf16 outputs; statistics differ only in summation order
    assert torch.equal(a, ops.gn_silu(out, gamma, beta, 32, 1e-5, True, tile_stats=stats))  # deterministic
    with pytest.raises(Exception):
        ops.conv_igemm(x[:, :100].contiguous(), wp, bias, N, 10, 10, 9, want_stats=True)  # H*W % 128 != 0


@pytest.mark.parametrize("cfg", ["1", "2", "3", "4"])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 7, 128, 320), (3, 32, 32, 320, 640), (1, 16, 16, 64, 48), (2, 16, 24, 192, 1280), (3, 16, 8, 64, 320)])
def test_conv3x3_on_the_dma_gemm_core_all_tilings(cfg, N, H, W, Cin, Cout, tune):
    """cd360_conv3x3_dma_bf16 (gemm8p.hip EPI 5: implicit im2col through LDS-DMA, padding through the buffer range check) in each of its
    four tilings (CD360_CONV_CFG: 256 x 320, 256 x 128, 256 x 256, 128 x 128) against torch's fp32 conv2d and against the register-
    staged kernel: bias + per-image addend + residual, ragged pixel / channel tiles, tiles straddling images, the per-slab channel
    statistics for the GroupNorm that follows ((3, 16, 8, ...): 384 pixels, i.e. a last pixel tile whose upper waves have no slab
    and must not write one)."""
    from cd360 import ops
    if cfg == "1" and Cout % 320:
        pytest.skip("the 320-channel tiling needs Cout % 320 == 0")
    tune(conv_cfg=int(cfg))
    g = torch.Generator().manual_seed(N * 100 + H + Cin + Cout)
    x = bf(torch.randn(N, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    emb = bf(torch.randn(N, Cout, generator=g))
    res = bf(torch.randn(N, Cout, H, W, generator=g))
    want = torch.nn.functional.conv2d(x, w, bias, padding=1) + emb[:, :, None, None] + res
    xt = x.permute(0, 2, 3, 1).reshape(N, H * W, Cin).contiguous().to(DEV, torch.bfloat16)
    wp = ops.pack_conv_weight(w).to(DEV)
    rt = res.permute(0, 2, 3, 1).reshape(N, H * W, Cout).contiguous().to(DEV, torch.bfloat16)
    args = (xt, wp, bias.to(DEV), N, H, W, 9, emb.to(DEV, torch.bfloat16), rt)
    got = ops.conv_igemm(*args)
    assert rel(got.reshape(N, H, W, Cout).permute(0, 3, 1, 2), want) < 8e-3
    if (H * W) % 128 == 0:
        out, stats = ops.conv_igemm(*args, want_stats=True)
        assert torch.equal(out, got)
        slabs = stats.shape[1]
        ref = out.float().reshape(N, slabs, H * W // slabs, Cout)
        assert rel(stats[..., 0], ref.sum(2)) < 1e-5 and rel(stats[..., 1], (ref * ref).sum(2)) < 1e-5
        assert torch.isfinite(stats).all()
    tune(conv_dma=0)
    assert rel(got, ops.conv_igemm(*args)) < 4e-3  # same sums in another order, both rounded to bf16


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 32, 32, 320, 640), (2, 64, 64, 128, 128), (1, 16, 16, 64, 48), (3, 16, 8, 64, 320), (2, 32, 32, 1280, 256),
                                            (1, 64, 64, 640, 80), (2, 8, 16, 192, 128)])
def test_conv3x3_halo_form_equals_the_tap_shifted_form(N, H, W, Cin, Cout, tune):
    """The halo form of the 3 x 3 convolution (gemm8p.hip EPI 11, round 6: the tile's input pixels fetched ONCE per 64-channel chunk into a
    (rows + 2) x (W + 2) image in the LDS, the nine taps read as row-shifted views of it, K walked chunk-major over weights that stay in
    cd360_conv_k_order's layout) against torch's fp32 conv2d and against the tap-shifted form (EPI 5, nine DMA passes): bias + per-image
    addend + residual, the GroupNorm slab statistics, image borders (every tile touches the left / right border; (1, 16, 16, ...) and
    (2, 8, 16, ...) have tiles that are a whole image or more than one image row band), K-split (Cin >= 384) and un-split arrangements,
    ragged channel tiles (48 / 80 / 320 of 128)."""
    from cd360 import ops
    g = torch.Generator().manual_seed(N * 100 + H + Cin + Cout)
    x = bf(torch.randn(N, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    emb = bf(torch.randn(N, Cout, generator=g))
    res = bf(torch.randn(N, Cout, H, W, generator=g))
    want = torch.nn.functional.conv2d(x, w, bias, padding=1) + emb[:, :, None, None] + res
    xt = x.permute(0, 2, 3, 1).reshape(N, H * W, Cin).contiguous().to(DEV, torch.bfloat16)
    wp = ops.pack_conv_weight(w).to(DEV)
    rt = res.permute(0, 2, 3, 1).reshape(N, H * W, Cout).contiguous().to(DEV, torch.bfloat16)
    args = (xt, wp, bias.to(DEV), N, H, W, 9, emb.to(DEV, torch.bfloat16), rt)
    tune(conv_halo=0, conv_cfg=4)
    shifted = ops.conv_igemm(*args)
    tune(conv_halo=1, conv_cfg=4)
    outs = [ops.conv_igemm(*args, want_stats=True) for _ in range(6)]
    got, stats = outs[0]
    assert rel(got.reshape(N, H, W, Cout).permute(0, 3, 1, 2), want) < 8e-3
    assert rel(got, shifted) < 4e-3  # the same products summed chunk-major instead of group-major, both rounded to bf16
    assert not torch.equal(got, shifted) or Cin == 64  # (it IS another kernel: one chunk has one summation order)
    slabs = stats.shape[1]
    ref = got.float().reshape(N, slabs, H * W // slabs, Cout)
    assert rel(stats[..., 0], ref.sum(2)) < 1e-5 and rel(stats[..., 1], (ref * ref).sum(2)) < 1e-5
    for o, st in outs[1:]:
        assert torch.equal(o, got) and torch.equal(st, stats)  # no ordering freedom: repeated launches agree bit for bit
    assert torch.equal(ops.conv_igemm(xt, wp, None, N, H, W, 9), ops.conv_igemm(xt, wp, torch.zeros(Cout, device=DEV), N, H, W, 9))


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 32, 32, 1280, 1280), (3, 64, 64, 640, 640), (2, 8, 8, 64, 64), (1, 5, 7, 128, 320), (2, 16, 12, 192, 80)])
def test_upsample_nearest2x_folded_into_the_convolution(N, H, W, Cin, Cout, tune):
    """Upsample.forward (openaimodel.py:114-181): nearest 2x + conv3x3 as ONE launch of four 2 x 2-tap phase convolutions of the source image
    (cd360_conv_up2x_bf16) against torch's fp32 interpolate + conv2d on the same bf16 values, and against the un-folded HIP path (the
    interpolated image through the 3 x 3 kernel); ragged / odd image sizes, every tiling the launch heuristic can choose."""
    from cd360 import ops
    g = torch.Generator().manual_seed(N * 100 + H + Cin + Cout)
    x = bf(torch.randn(N, Cin, H, W, generator=g))
    w = bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(Cout, generator=g)
    want = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"), w, bias, padding=1)
    xt = x.permute(0, 2, 3, 1).reshape(N, H * W, Cin).to(DEV, torch.bfloat16).contiguous()
    wp = ops.pack_upsample_conv_weight(w.to(DEV))
    cfgs = [None] + [c for c in (2, 3, 4) if True]
    for cfg in cfgs:
        tune(conv_cfg=-1 if cfg is None else cfg)
        got = ops.conv_up2x(xt, wp, bias.to(DEV), N, H, W).reshape(N, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
        assert rel(got, want) < 1e-2, cfg
        assert torch.equal(got, ops.conv_up2x(xt, wp, bias.to(DEV), N, H, W).reshape(N, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2))
    tune(conv_cfg=-1)
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest").permute(0, 2, 3, 1).reshape(N, 4 * H * W, Cin).to(DEV, torch.bfloat16).contiguous()
    plain = ops.conv_igemm(up, ops.pack_conv_weight(w.to(DEV)), bias.to(DEV), N, 2 * H, 2 * W, 9).reshape(N, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2)
    assert rel(got, plain) < 8e-3  # same sums with the coinciding taps' weights added before the bf16 rounding instead of after


@pytest.mark.parametrize("N,H,W,Cin,Cout,stride", [(2, 16, 16, 64, 64, 2), (3, 32, 32, 320, 320, 2), (1, 8, 12, 128, 160, 2), (2, 16, 16, 4, 320, 1), (2, 16, 16, 320, 4, 1)])
def test_conv_stride2_and_padded_channels_through_the_module_wrapper(N, H, W, Cin, Cout, stride):
    """Downsample.op (conv3x3 stride 2 pad 1, openaimodel.py:190-213) and the UNet's 4 -> 320 / 320 -> 4 convs (:663-670,967-973)
    on the implicit-GEMM kernel: conv_image(nn.Conv2d, x) vs torch's fp32 conv2d on the same bf16 inputs."""
    from sgm.modules.diffusionmodules.util import conv_image, packed_conv
    g = torch.Generator().manual_seed(Cin * 7 + Cout + stride)
    conv = torch.nn.Conv2d(Cin, Cout, 3, stride=stride, padding=1)
    with torch.no_grad():
        conv.weight.copy_(bf(torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5))
        conv.bias.copy_(bf(torch.randn(Cout, generator=g)))
    x = bf(torch.randn(N, Cin, H, W, generator=g))
    want = torch.nn.functional.conv2d(x, conv.weight.float(), conv.bias.float(), stride=stride, padding=1)
    conv = conv.to(DEV, torch.bfloat16)
    assert packed_conv(conv) is not None
    with torch.no_grad():
        got = conv_image(conv, x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert got.shape == want.shape and rel(got, want) < 8e-3


# ------------------------------------------------------------------------------------------------ GroupNorm + SiLU (K7)
@pytest.mark.parametrize("N,P,C,silu", [(2, 64, 64, True), (3, 1024, 320, True), (1, 4096, 640, False), (2, 256, 2560, True), (1, 100, 960, False)])
def test_gn_silu(N, P, C, silu):
    from cd360 import ops
    g = torch.Generator().manual_seed(C + P)
    x = bf(torch.randn(N, P, C, generator=g) * 2 + 0.5)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    want = torch.nn.functional.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    if silu:
        want = torch.nn.functional.silu(want)
    got = ops.gn_silu(x.to(DEV, torch.bfloat16), gamma.to(DEV), beta.to(DEV), 32, 1e-5, silu)
    assert rel(got, want) < 8e-3


# ------------------------------------------------------------------------------------------------ fp8-MFMA cross-attention (configs[4])
@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 3, 200, 77), (2, 2, 96, 40), (3, 10, 4096, 77), (1, 10, 98304, 77)])
def test_attention_fp8_mfma_variant_tolerance(B, H, Nq, Nk):
    """BASELINE configs[4]: QK^T and PV on e4m3 MFMA, bf16 tensors in and out, up to the shape the config is quoted for (the 98 304 pose
    tokens of one batch element at 1024^2).  Stated tolerance on unit-normal q / k / v with per-tensor amax / 448 scales (measured 5e-2
    RMS, 6e-2 - 1e-1 max-norm at SDXL shapes, profiles/r03_fp8_tolerance.json): RMS < 7e-2 and max-norm < 1.5e-1 against the fp32 oracle
    -- six times outside the 1e-2 bar the bf16 kernel meets and 4 % slower at the pose-token shape, which is why the variant is retired
    from the product path and kept as this measured record.  NaN padding beyond Nk must never leak; Nk > 96 is refused."""
    from cd360 import ops
    g = torch.Generator().manual_seed(B * 1000 + Nq + Nk)
    q = bf(torch.randn(B, Nq, H * 64, generator=g))
    k = bf(torch.randn(B, Nk, H * 64, generator=g))
    v = bf(torch.randn(B, Nk, H * 64, generator=g))
    nkp = (Nk + 7) // 8 * 8
    vp = torch.full((B, nkp, H * 64), float("nan"))
    vp[:, :Nk] = v
    kp = torch.full((B, nkp, H * 64), float("nan"))
    kp[:, :Nk] = k
    args = (q.to(DEV, torch.bfloat16), kp.to(DEV, torch.bfloat16), vp.to(DEV, torch.bfloat16), H)
    o8 = ops.attention_fp8mfma(*args, nk=Nk)
    o16 = ops.attention(*args, nk=Nk)

    def split(t):
        return t.reshape(B, t.shape[1], H, 64).permute(0, 2, 1, 3).reshape(B * H, t.shape[1], 64)

    want = O.attention_core(split(q), split(k), split(v)).reshape(B, H, Nq, 64).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    rms = lambda a, b_: float(((a.float().cpu() - b_.float().cpu()) ** 2).mean().sqrt() / (b_.float().cpu() ** 2).mean().sqrt())
    assert rel(o16, want) < 1e-2
    assert rel(o8, want) < 1.5e-1 and rms(o8, want) < 7e-2, (rel(o8, want), rms(o8, want))
    assert rms(o8, want) > 5 * rms(o16, want)  # the variant really runs in fp8 (bf16 is ~2e-3)
    big = torch.zeros(1, 104, 64, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(Exception):
        ops.attention_fp8mfma(big, big, big, 1)  # Nk = 104 > 96


def test_pose_embed_c_abi():
    """A11 through the C ABI: pose_emb_layers(cat[x, xref]) == x Wa^T + xref Wb^T (attention.py:515-516,634)."""
    from cd360 import ops
    g = torch.Generator().manual_seed(17)
    for rows, C in ((300, 640), (1024, 1280), (77, 64)):
        x, xr = bf(torch.randn(rows, C, generator=g)), bf(torch.randn(rows, C, generator=g))
        w = bf(torch.cat([torch.eye(C), torch.zeros(C, C)], 1) + 0.05 * torch.randn(C, 2 * C, generator=g))
        want = torch.cat([x, xr], -1) @ w.t()
        d = lambda t: t.to(DEV, torch.bfloat16).contiguous()
        got = ops.pose_embed(d(x), d(xr), d(w[:, :C]), d(w[:, C:]))
        assert rel(got, want) < 1e-2


@pytest.mark.parametrize("C", [64, 640, 1280])
def test_rowdot1(C):
    """cd360_rowdot1_bf16 (lv = xref . vf of the reference tables, cd360/nerf.py reference_tables) vs the fp32 product"""
    from cd360 import ops
    torch.manual_seed(3)
    h = torch.randn(3, 1000, C).to(torch.bfloat16).float()
    w = torch.randn(C)
    got = ops.rowdot1(h.to(DEV, torch.bfloat16), w.to(DEV))
    assert got.shape == (3, 1000) and rel(got, h @ w) < 1e-5
    assert torch.equal(got, ops.rowdot1(h.to(DEV, torch.bfloat16), w.to(DEV)))


@pytest.mark.parametrize("S,N", [(24, 24), (32, 32), (8, 20)])
def test_sample_pdf(S, N):
    """cd360_sample_pdf (pytorch3d._C.sample_pdf's place at nerfsd_pytorch3d.py:300-305; f4) against the oracle's inverse-CDF sampler:
    random, empty and single-bin weight rows, u at both ends, the in-place form and the gaps."""
    from cd360 import ops
    g = torch.Generator().manual_seed(S)
    rows = 4099
    bins = (torch.linspace(0.5, 3.0, S + 1)[None] + 0.01 * torch.rand(rows, S + 1, generator=g)).sort(-1).values
    w = torch.rand(rows, S, generator=g) ** 3
    w[5] = 0.0
    w[6] = 0.0
    w[6, S // 2] = 2.0
    u = torch.rand(rows, N, generator=g)
    u[7, 0], u[7, -1] = 0.0, 0.999999
    u[8] = u[8].sort().values
    want = O.sample_pdf(bins, w, u, 1e-5)
    got, dists = ops.sample_pdf(bins.to(DEV), w.to(DEV), u.to(DEV), 1e-5, want_dists=True)
    # conditioning: inside a bin of probability mass p the sample moves by (bin width) x (cdf rounding, ~1e-7) / p, so bins that barely clear
    # eps amplify the last bit of the cdf (a sequential fp32 sum here, torch's blocked sum in the oracle) to ~1e-4: the agreement
    # asked scales with 1 / p -- and everywhere the forward map must hold: F(sample) = u for the float64 cdf F
    wdd = (w + 1e-5).double()
    cdf = torch.cat([torch.zeros(rows, 1, dtype=torch.float64), torch.cumsum(wdd / wdd.sum(-1, keepdim=True), -1)], -1)
    k = (torch.searchsorted(cdf, u.double().contiguous(), right=True) - 1).clamp(0, S - 1)
    mass = torch.gather(cdf, -1, k + 1) - torch.gather(cdf, -1, k)
    err = (got.cpu() - want).abs()
    assert (err <= 2e-6 + 1e-7 / mass.clamp_min(1e-5).float()).all()  # bin width ~0.1 x (cdf error <~ 1e-6) / mass
    F = torch.from_numpy(np.stack([np.interp(got[i].double().cpu().numpy(), bins[i].double().numpy(), cdf[i].numpy()) for i in range(0, rows, 37)]))
    assert (F - u[::37].double()).abs().max() < 2e-5  # eps-sized steps where a bin's mass is below eps (the `denom = 1` rule)
    gc = got.cpu()
    wd = torch.cat([gc[:, 1:] - gc[:, :-1], bins[:, -1:] - gc[:, -1:]], -1)
    assert (dists.cpu() - wd).abs().max() < 1e-6
    ud = u.to(DEV).clone()
    assert ops.sample_pdf(bins.to(DEV), w.to(DEV), ud, 1e-5, inplace=True) is ud and torch.equal(ud, got)
    assert ((got >= bins[:, :1].to(DEV)) & (got <= bins[:, -1:].to(DEV))).all()
    srt = ops.sample_pdf(bins.to(DEV), w.to(DEV), u.sort(-1).values.to(DEV))
    assert (srt[:, 1:] >= srt[:, :-1]).all()


@pytest.mark.parametrize("N,H,W,Cin", [(3, 128, 128, 320), (2, 64, 64, 128), (1, 32, 32, 64), (2, 6, 32, 192)])
def test_out_conv4_matches_conv2d(N, H, W, Cin):
    """cd360_out_conv4_bf16 (the UNet's 320 -> 4 output convolution as a [pixels x 36] x Cin MFMA product over each two-row band and its halo
    rows + nine shifted fp32 adds) against torch's fp32 conv2d on the same bf16 values: borders (zero padding on all four sides), every
    band position (first / last band read a halo row outside the image), image widths 32 / 64 / 128, repeated launches bit-identical."""
    from cd360 import ops
    g = torch.Generator().manual_seed(N * 1000 + H + Cin)
    x = bf(torch.randn(N, Cin, H, W, generator=g))
    w = bf(torch.randn(4, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5)
    bias = torch.randn(4, generator=g)
    want = torch.nn.functional.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1).reshape(N, H * W, 4)
    xt = x.permute(0, 2, 3, 1).reshape(N, H * W, Cin).contiguous().to(DEV, torch.bfloat16)
    wp = ops.pack_out_conv4_weight(w.to(DEV, torch.bfloat16))
    got = ops.out_conv4(xt, wp, bias.to(DEV), N, H, W)
    assert got.shape == (N, H * W, 4) and rel(got, want) < 6e-3
    for _ in range(4):
        assert torch.equal(ops.out_conv4(xt, wp, bias.to(DEV), N, H, W), got)
    # against the implicit-GEMM kernel with Cout padded to 16 (the round-5 path): the same products in another order
    wpad = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, 12))
    old = ops.conv_igemm(xt, ops.pack_conv_weight(wpad).to(DEV), torch.nn.functional.pad(bias, (0, 12)).to(DEV), N, H, W, 9)[..., :4]
    assert rel(got, old) < 6e-3
