"""BASELINE configs[2] on the GPU box: a batch of 8 target poses through the sampling job of cd360/job.py -- shard.assign_poses ->
ONE captured Sampler retargeted pose after pose -> shard.gather_latents -- on the one GPU the box has, and `bench.py --gpus N` launching
its own ranks.  The reference loops its poses sequentially on one GPU (sample.py:331-349).  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _job(p, latent, refs):
    from cd360 import synth
    one = [synth.pose_batch(1, refs, seed=100 + p, n_train=50)[0]]
    g = torch.Generator(device=DEV).manual_seed(7 + p)
    ctx = torch.randn(3, 77, 2048, generator=g, device=DEV).to(BF)
    y = torch.randn(3, 2816, generator=g, device=DEV).to(BF)
    return (one * 3, ctx, y, torch.randn(1, 4, latent, latent, generator=g, device=DEV))


@torch.no_grad()
def test_eight_poses_through_shard_retarget_gather_equal_eight_fresh_samplers_and_the_oracle():
    """8 target poses, world 1: the rank's share is all 8 (assign_poses(8, 1, 0)), ONE graph-mode Sampler walks them through
    Sampler.retarget x 7 (the packed camera buffer and the CFG conditioning rewritten in place under the two captured hipGraphs), 3 denoise
    steps each (render + 2 cached), gather_latents returns [8, 4, L, L].
    (1) every pose's latent is BIT-IDENTICAL to a fresh graph-mode single-pose sampler's at that pose: nothing of an earlier pose -- camera
        packing, memoised camera constants, context K / V, cached render -- survives a retarget;
    (2) the render the retargeted sampler holds after the LAST pose (seven retargets deep) and the one a fresh sampler holds after the FIRST
        are the oracle's: the 640-level pose block's rendered features for all three CFG branches against oracle.reference_attn's chain
        (FeatureNeRF -> pose-token cross-attention -> volume render) on every ray, 1e-2 of the tensor maximum."""
    import bench
    from cd360 import job, sampling, shard
    from cd360.cameras import pack_cameras
    from test_modules_gpu import _oracle_render_on_rays, rel
    latent, refs, steps, P = 32, 6, 3, 8
    net = bench.build_model(latent, refs, 50, DEV)
    assert shard.assign_poses(P, 1, 0) == list(range(P))
    name0, blk0 = sampling.pose_blocks(net)[0]
    held = {}

    def make_sampler(pose, ctx, y):
        held["smp"] = job.Sampler(net, pose, ctx, y, 50, use_graph=True)
        return held["smp"]

    latents, mine = job.sample_poses(make_sampler, lambda p: _job(p, latent, refs), P, steps, world=1, rank=0)
    assert mine == list(range(P)) and latents.shape == (P, 4, latent, latent) and torch.isfinite(latents).all()
    assert held["smp"].rgraph is not None, "the render step must have been captured: the retargets are then replays"
    rend_last = blk0.rendered_feat.float().clone()  # pose 7's render, produced by a replay of the graph captured at pose 0

    rend_first = None
    for p in range(P):
        pose, ctx, y, x0 = _job(p, latent, refs)
        fresh = job.sample_assigned(job.Sampler(net, pose, ctx, y, 50, use_graph=True), [(pose, ctx, y, x0)], steps)[0]
        assert torch.equal(fresh, latents[p:p + 1]), (p, float((fresh - latents[p:p + 1]).abs().max()))
        if p == 0:
            rend_first = blk0.rendered_feat.float().clone()
    assert float((latents[0] - latents[1]).abs().max() / latents[1].abs().max()) > 1e-2  # different trajectories

    # ---- (2) oracle: the first pose block (input_blocks.4.1.transformer_blocks.0: C = 640, 10 heads, r = latent / 2) ----
    w = {k: v.detach().float().cpu() for k, v in blk0.state_dict().items() if "references" not in k and "raymarcher" not in k}
    allrefs = blk0.references.float().cpu()
    choices = list(blk0.reference_choices)
    hw = allrefs.shape[1]
    idx = torch.arange(hw)
    cond = allrefs[:-1][torch.tensor(choices)][None]
    null = allrefs[-1:][None].expand(1, len(choices), -1, -1)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    for p, rend in ((0, rend_first), (P - 1, rend_last)):
        pose, ctx, y, _ = _job(p, latent, refs)
        cams = pack_cameras(pose[:1]).float().cpu()
        # the guider's batch is [uc | uc | c] with the null IMAGE for the first third (guiders.py:102-133, sample.py:89-96)
        smp_ctx = job.Sampler(net, pose, ctx, y, 50, use_graph=False).ctx.float().cpu()
        errs = []
        for br in range(3):
            want = _oracle_render_on_rays(w, cams, null if br == 0 else cond, smp_ctx[br:br + 1], blk0.attn2.heads,
                                          blk0.pose_featurenerf.num_samples, float(blk0.pose_featurenerf.far), idx)
            errs.append(rel(rend[br:br + 1], want[0]))
        print(f"pose {p}: rendered features of {name0} vs oracle per CFG branch:", [round(e, 5) for e in errs])
        assert max(errs) < 1e-2, (p, errs)


@torch.no_grad()
def test_eight_poses_at_the_headline_size_on_one_gpu():
    """BASELINE configs[2] at FULL size on the one GPU of the box (round-5 review, weak #2): 8 target poses x (render + 2 cached steps) at
    latent 128^2 with 50 reference views through job.sample_poses -- ONE captured Sampler, seven retargets (each re-renders all 12 pose
    blocks at 1024^2 from the rewritten camera buffer).
    (1) the latents of pose 0 and of pose 7 (seven retargets deep) are BIT-IDENTICAL to fresh graph-mode samplers at those poses;
    (2) the level-1 render (C = 640, r = 64, 98 304 pose tokens per CFG branch) the retargeted sampler holds after pose 7 is the oracle's
        on 64 rays spread over the image incl. its four corners, all three CFG branches, 1e-2 of the tensor maximum."""
    import bench
    import numpy as np
    from cd360 import job, sampling, shard
    from cd360.cameras import pack_cameras
    from test_modules_gpu import _oracle_render_on_rays, rel
    latent, refs, steps, P = 128, 50, 3, 8
    net = bench.build_model(latent, refs, 50, DEV)
    name0, blk0 = sampling.pose_blocks(net)[0]
    held = {}

    def make_sampler(pose, ctx, y):
        held["smp"] = job.Sampler(net, pose, ctx, y, 50, use_graph=True)
        return held["smp"]

    latents, mine = job.sample_poses(make_sampler, lambda p: _job(p, latent, refs), P, steps, world=1, rank=0)
    assert mine == list(range(P)) and latents.shape == (P, 4, latent, latent) and torch.isfinite(latents).all()
    assert held["smp"].rgraph is not None and held["smp"].staged
    rend_last = blk0.rendered_feat.float().clone()
    del held["smp"]
    for p in (0, P - 1):
        pose, ctx, y, x0 = _job(p, latent, refs)
        fresh = job.sample_assigned(job.Sampler(net, pose, ctx, y, 50, use_graph=True), [(pose, ctx, y, x0)], steps)[0]
        assert torch.equal(fresh, latents[p:p + 1]), (p, float((fresh - latents[p:p + 1]).abs().max()))
    assert float((latents[0] - latents[P - 1]).abs().max() / latents[P - 1].abs().max()) > 1e-2
    # ---- the oracle on a ray subset of pose 7's render ----
    w = {k: v.detach().float().cpu() for k, v in blk0.state_dict().items() if "references" not in k and "raymarcher" not in k}
    allrefs = blk0.references.float().cpu()
    choices = list(blk0.reference_choices)
    hw = allrefs.shape[1]
    r = int(round(hw ** 0.5))
    g = np.random.default_rng(5)
    idx = torch.tensor(sorted(set([0, r - 1, hw - r, hw - 1]) | set(int(v) for v in g.choice(hw, 60, replace=False)))[:64])
    cond = allrefs[:-1][torch.tensor(choices)][None]
    null = allrefs[-1:][None].expand(1, len(choices), -1, -1)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    pose, ctx, y, _ = _job(P - 1, latent, refs)
    cams = pack_cameras(pose[:1]).float().cpu()
    smp_ctx = job.Sampler(net, pose, ctx, y, 50, use_graph=False).ctx.float().cpu()
    errs = []
    for br in range(3):
        want = _oracle_render_on_rays(w, cams, null if br == 0 else cond, smp_ctx[br:br + 1], blk0.attn2.heads,
                                      blk0.pose_featurenerf.num_samples, float(blk0.pose_featurenerf.far), idx)
        errs.append(rel(rend_last[br:br + 1, idx], want[0]))
    print(f"pose {P - 1} at latent 128 / 50 views: rendered features of {name0} vs oracle on {len(idx)} rays per CFG branch:", [round(e, 5) for e in errs])
    assert max(errs) < 1e-2, errs


def test_bench_gpus_2_starts_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus 2 ...` with no launcher around it (no WORLD_SIZE): bench.py re-executes itself under torch.distributed.run,
    two ranks share the box's one GPU over gloo (CD360_BENCH_ONE_GPU: a control-path sanity run, never a measurement -- RCCL refuses two
    ranks on one device), 4 poses are sharded 2 + 2, each rank retargets once, the final latents are all-gathered, rank 0 prints ONE JSON
    line with n_gpus = 2."""
    env = dict(os.environ, CD360_BENCH_ONE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--poses", "4", "--steps", "2", "--warmup", "0", "--latent", "32", "--refs", "6",
           "--no-train-step", "--no-cpu-baseline", "--no-profile"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["config"]["poses"] == 4 and line["config"]["poses_per_gpu"] == 2
    assert line["scaling"] == "weak" and line["value"] > 0 and "sanity_run" in line["config"]
    assert set(line["config"]["rank_ms_per_step"]) == {"min", "max"}
