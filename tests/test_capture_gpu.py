"""hipGraph capture against the per-forward memo tables (cd360/memo.py).  Needs an MI355X.

bench.py's graph mode runs one eager render, captures the render step, and later points the captured sampler at the next target pose by
rewriting the packed camera tensor IN PLACE (Sampler.retarget).  A value memoised by the eager run under (tensor, version) must not be
served to the capture: its kernels would be missing from the graph and every replay after a retarget would read the old pose's value."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


@torch.no_grad()
def test_memoised_values_of_an_eager_run_never_feed_a_capture():
    from cd360 import nerf, synth
    from cd360.cameras import pack_cameras
    from sgm.modules.attention import _pad_tokens
    cams = pack_cameras(synth.pose_batch(2, 5, seed=1)).to(DEV)
    other = pack_cameras(synth.pose_batch(2, 5, seed=2)).to(DEV)
    xs, ys = nerf.patch_positions(8, DEV), nerf.patch_positions(8, DEV)
    ctx = torch.randn(2, 77, 64, device=DEV).to(torch.bfloat16)
    ctx2 = torch.randn(2, 77, 64, device=DEV).to(torch.bfloat16)

    def step():  # what every pose block of a forward asks for
        return nerf._camera_constants(cams) + 0.0, nerf._plucker_rows(cams, xs, ys).float() + 0.0, _pad_tokens(ctx).float() + 0.0

    eager = [t.clone() for t in step()]  # fills the memo tables under (tensor, version)
    g, outs = _capture(step)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, eager):
        assert torch.equal(a, b)
    cams.copy_(other)  # Sampler.retarget: the next pose, written into the buffer the graph reads
    ctx.copy_(ctx2)
    g.replay()
    torch.cuda.synchronize()
    fresh = step()  # eager, new versions: recomputed
    for a, b, old in zip(outs, fresh, eager):
        assert torch.equal(a, b), "the replay must follow the rewritten buffers"
        assert not torch.equal(b, old)
    # within one eager forward the tables still serve their purpose: the same object comes back
    assert nerf._camera_constants(cams) is nerf._camera_constants(cams)
    assert nerf._plucker_rows(cams, xs, ys) is nerf._plucker_rows(cams, xs, ys)
    assert _pad_tokens(ctx) is _pad_tokens(ctx)


@pytest.mark.parametrize("fp8", [False, True])
@torch.no_grad()
def test_graph_mode_sampler_follows_a_retarget_like_the_eager_one(fp8):
    """bench.py's Sampler end to end at the SDXL UNet (small latent): two target poses sampled one after the other by ONE sampler -- the
    second through Sampler.retarget -- replayed from the captured render / steady graphs, against the same two trajectories launched
    eagerly.  Every value a captured step derives from the pose or the conditioning must be re-derived inside the graph."""
    sys.path.insert(0, ROOT)
    import bench
    from cd360 import synth
    latent, refs, steps = 32, 6, 3
    net = bench.build_model(latent, refs, 50, DEV)
    jobs = []
    for pi in (0, 1):
        one = [synth.pose_batch(1, refs, seed=100 + pi, n_train=50)[0]]
        g = torch.Generator(device=DEV).manual_seed(7 + pi)
        ctx = torch.randn(3, 77, 2048, generator=g, device=DEV).to(torch.bfloat16)
        y = torch.randn(3, 2816, generator=g, device=DEV).to(torch.bfloat16)
        jobs.append((one * 3, ctx, y, torch.randn(1, 4, latent, latent, generator=g, device=DEV)))

    def run(use_graph):
        smp = bench.Sampler(net, *jobs[0][:3], 50, use_graph=use_graph)
        outs = []
        for j, (pose, ctx, y, x) in enumerate(jobs):
            if j > 0:
                smp.retarget(pose, ctx, y)
            xs = x.clone()
            for i in range(steps):
                xs = smp.step(xs, i)
            outs.append(xs.clone())
        return outs

    from cd360 import routes
    with routes.override(fp8_attn=fp8):  # BASELINE configs[4]: the e4m3 image of the context K / V is one more pinned, re-packed buffer
        eager = run(False)
        graph = run(True)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    assert rel(eager[1], eager[0]) > 1e-2  # two different trajectories
    for e, g_ in zip(eager, graph):
        assert torch.isfinite(g_).all() and rel(g_, e) < 2e-3, rel(g_, e)


@torch.no_grad()
def test_two_poses_per_replay_equal_two_single_pose_trajectories():
    """bench.py --poses-per-replay 2: two target poses batched into one denoise step (CFG batch 6 = [null x2 | image x2 | image+text x2],
    de-duplicated render of four elements) against the same two poses sampled one by one, replayed from hipGraphs."""
    sys.path.insert(0, ROOT)
    import bench
    from cd360 import synth
    latent, refs, steps = 32, 6, 3
    net = bench.build_model(latent, refs, 50, DEV)
    singles = []
    for pi in (0, 1):
        one = [synth.pose_batch(1, refs, seed=100 + pi, n_train=50)[0]]
        g = torch.Generator(device=DEV).manual_seed(7 + pi)
        ctx = torch.randn(3, 77, 2048, generator=g, device=DEV).to(torch.bfloat16)
        y = torch.randn(3, 2816, generator=g, device=DEV).to(torch.bfloat16)
        singles.append((one * 3, ctx, y, torch.randn(1, 4, latent, latent, generator=g, device=DEV)))
    pose2 = [singles[0][0][0], singles[1][0][0]] * 3
    ctx2 = torch.cat([torch.cat([singles[0][1][k:k + 1], singles[1][1][k:k + 1]]) for k in range(3)])
    y2 = torch.cat([torch.cat([singles[0][2][k:k + 1], singles[1][2][k:k + 1]]) for k in range(3)])

    def sample(pose, ctx, y, x):
        smp = bench.Sampler(net, pose, ctx, y, 50, use_graph=True)
        for i in range(steps):
            x = smp.step(x, i)
        return x.clone()

    outs = [sample(*job[:3], job[3].clone()) for job in singles]
    both = sample(pose2, ctx2, y2, torch.cat([singles[0][3], singles[1][3]]))
    for k in range(2):  # the level-2 GEMMs take other tiles at M = 6144 than at 3072: same products, other blocking -- rounding only
        assert float((both[k:k + 1] - outs[k]).abs().max() / outs[k].abs().max()) < 5e-3
    assert float((outs[0] - outs[1]).abs().max() / outs[1].abs().max()) > 1e-2


@torch.no_grad()
def test_two_streams_hold_their_own_tuning_and_prefetch_arm():
    """SURVEY.md section 8b "re-entrant across streams": a tiling override set for ONE stream (cd360_set_stream_tuning) is read by the
    launches issued on that stream and by nothing else -- the shape query that sizes the row-statistics buffer follows it
    (cd360_query_stream), the default and the other stream keep the shape-chosen tiling -- and a weight-prefetch arm of one capturing
    stream (cd360_prefetch_arm_on) adds touch kernels to THAT capture only.  Same products either way: results agree to bf16 round-off."""
    from cd360 import _lib, ops
    from cd360.prefetch import WeightPrefetcher
    g = torch.Generator(device=DEV).manual_seed(0)
    M, N, K = 3072, 1280, 1280
    a = torch.randn(M, K, generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=DEV) * K ** -0.5).to(torch.bfloat16)
    res = torch.randn(M, N, generator=g, device=DEV).to(torch.bfloat16)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    base = _lib.get_tuning()
    default_tile = ops.gemm_tile_n(M, N)
    forced = 3 if default_tile != 256 else 2  # gemm_cfg 3 = 256 x 256 tiles, 2 = 128 x 128: whichever the shape does NOT pick
    _lib.set_stream_tuning(s1, gemm_cfg=forced)
    try:
        assert _lib.get_tuning() == base and _lib.get_stream_tuning(s2) == base and _lib.get_stream_tuning(s1)["gemm_cfg"] == forced
        outs, parts = {}, {}
        torch.cuda.synchronize()
        for name, st in (("s1", s1), ("s2", s2)):
            with torch.cuda.stream(st):
                parts[name] = ops.gemm_tile_n(M, N)
                o, stats = ops.gemm(a, w, res=res, want_stats=True)
                outs[name] = (o, stats)
        torch.cuda.synchronize()
        assert parts["s2"] == default_tile and parts["s1"] != default_tile, parts
        assert outs["s1"][1].shape[1] != outs["s2"][1].shape[1]  # the statistics buffer was sized for the tiling its stream launched
        rel = lambda x, y: float((x.float() - y.float()).abs().max() / y.float().abs().max())
        assert rel(outs["s1"][0], outs["s2"][0]) < 1e-2 and not torch.equal(outs["s1"][1].sum(1), outs["s1"][1].sum(1) * 0)
        assert rel(outs["s1"][1].sum(1), outs["s2"][1].sum(1)) < 2e-2  # per-row (sum, sumsq) over all column tiles: same rows
        assert ops.gemm_tile_n(M, N) == default_tile  # back on the default stream: untouched
    finally:
        _lib.clear_stream_tuning(s1)
    assert _lib.get_stream_tuning(s1) == base
    # prefetch arms are per capturing stream too: s1 captures two launches WITH the prefetcher (armed for s1 only), s2 launches the same
    # product eagerly with nothing armed on it; the replayed capture and the eager launch agree bit for bit
    big_w = (torch.randn(10240, 1280, generator=g, device=DEV) * 0.03).to(torch.bfloat16)  # 26 MB: above the prefetcher's size floor
    x = torch.randn(3072, 1280, generator=g, device=DEV).to(torch.bfloat16)
    ops.gemm(x, big_w)
    torch.cuda.synchronize()
    pf = WeightPrefetcher(torch.device(DEV))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s1):
        with pf:
            assert pf.main == s1.cuda_stream
            y1 = ops.gemm(x, big_w)
            y1b = ops.gemm(y1[:, :1280].contiguous(), big_w)
    with torch.cuda.stream(s2):
        y2 = ops.gemm(x, big_w)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.isfinite(y1b.float()).all()
