"""The N>1 job structure on CPU: pose sharding + the single all-gather of final latents, world_size 2 and 3 over gloo."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_poses, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "custom-diffusion360_amd"))
    from cd360 import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.assign_poses(num_poses, world, rank)
    local = torch.stack([torch.full((4, 8, 8), float(p)) for p in mine]) if mine else torch.zeros(0, 4, 8, 8)
    allx = shard.gather_latents(local, num_poses)
    ok = allx.shape == (num_poses, 4, 8, 8) and all(torch.all(allx[p] == p) for p in range(num_poses))
    q.put((rank, mine, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,num_poses", [(2, 8), (3, 8), (2, 1)])
def test_pose_sharding_and_latent_allgather(world, num_poses):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_poses, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seen = [p for _, mine, _ in res for p in mine]
    assert seen == list(range(num_poses))  # every pose exactly once, contiguous blocks in rank order
    assert all(ok for _, _, ok in res)
    sizes = [len(m) for _, m, _ in res]
    assert max(sizes) - min(sizes) <= 1


def test_assign_poses_partition():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "custom-diffusion360_amd"))
    from cd360.shard import assign_poses
    for world in (1, 2, 4, 8):
        for P in (1, 7, 8, 50):
            parts = [assign_poses(P, world, r) for r in range(world)]
            assert sum(parts, []) == list(range(P))


class _Blk(torch.nn.Module):
    """Just enough of a pose block for the harvest: lives under `transformer_blocks.N` and has `pose_emb_layers`."""

    def __init__(self):
        super().__init__()
        self.pose_emb_layers = torch.nn.Linear(4, 2, bias=False)


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.transformer_blocks = torch.nn.ModuleList([_Blk(), torch.nn.Identity(), _Blk()])


def _harvest_worker(rank, world, port, n_images, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "custom-diffusion360_amd"))
    from cd360 import finetune
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _Net()
    # DistributedSampler order: reference image i is processed by rank i % world, in increasing i, one per forward
    acts = {}
    for name in ("transformer_blocks.0", "transformer_blocks.2"):
        acts[name] = [torch.full((1, 3, 2), float(i)) for i in range(rank, n_images, world)]
    refs = finetune.harvest_references(net, acts)
    ok = all(r.shape == (n_images, 3, 2) and all(torch.all(r[i] == i) for i in range(n_images)) for r in refs.values())
    ok = ok and torch.equal(net.transformer_blocks[2].references, refs["transformer_blocks.2"]) and len(refs) == 2
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_images", [(2, 6), (3, 6)])
def test_references_harvest_allgather_restores_dataset_order(world, n_images):
    """main.py:596-607: per-rank features -> all_gather -> transpose(0, 1) -> flatten gives rows in dataset order."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_harvest_worker, args=(r, world, port, n_images, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _dp_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "custom-diffusion360_amd"))
    from cd360 import finetune
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.ones(3, 5, dtype=torch.bfloat16)), torch.nn.Parameter(torch.ones(7, dtype=torch.bfloat16))]
    opt = finetune.MasterAdamW(params, lr=1e-2, weight_decay=0.0)
    for p in params:  # rank-dependent gradients: the average over ranks is (world + 1) / 2
        p.grad = torch.full_like(p, float(rank + 1))
    opt.allreduce_grads()
    avg_ok = all(torch.all(p.grad.float() == (world + 1) / 2) for p in params)
    opt.step()
    gathered = [torch.empty(22) for _ in range(world)]
    dist.all_gather(gathered, torch.cat([m.reshape(-1) for m in opt.master]))
    same = all(torch.equal(g, gathered[0]) for g in gathered)
    q.put((rank, bool(avg_ok), bool(same), float(opt.master[0][0, 0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_gradient_allreduce(world):
    """finetune.MasterAdamW.allreduce_grads: ONE flat all-reduce averages the trainable gradients over the ranks (the reference trains
    under DDP); afterwards every rank takes the identical AdamW step on its fp32 masters."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(avg_ok and same for _, avg_ok, same, _ in res)
    assert all(abs(v - (1.0 - 1e-2)) < 1e-6 for _, _, _, v in res)  # first AdamW step = -lr * sign(grad)


class _CpuSampler:
    """Stand-in with the Sampler protocol of cd360/job.py (retarget / step) on CPU tensors: a deterministic 'denoise step' that depends on
    the pose, the conditioning and the step index, so that a pose sampled on the wrong rank, with stale conditioning, or for the wrong number
    of steps gives a different latent."""

    def __init__(self, pose, ctx, y):
        self.retarget(pose, ctx, y)

    def retarget(self, pose, ctx, y):
        self.k = float(pose) + float(ctx.sum()) * 1e-3 + float(y.sum()) * 1e-4

    def step(self, x, i):
        return x * 0.5 + self.k + 0.01 * i


def _job_of(p):
    g = torch.Generator().manual_seed(50 + p)
    return (p, torch.randn(3, 7, generator=g), torch.randn(3, 5, generator=g), torch.randn(1, 4, 8, 8, generator=g))


def _job_worker(rank, world, port, num_poses, steps, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "custom-diffusion360_amd"))
    from cd360 import job
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    latents, mine = job.sample_poses(_CpuSampler, _job_of, num_poses, steps, world, rank)
    q.put((rank, mine, latents))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,num_poses", [(2, 8), (3, 8)])
def test_sample_poses_job_matches_one_fresh_sampler_per_pose(world, num_poses):
    """BASELINE configs[2] on CPU: cd360.job.sample_poses (assign_poses -> ONE sampler per rank, retargeted pose after pose -> one
    all-gather) over gloo.  Every rank ends with the latents of ALL poses in pose order, each bit-identical to a fresh single-pose
    sampler's (the same loop `bench.py` times and tests/test_job_gpu.py runs on the GPU with the captured Sampler)."""
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "custom-diffusion360_amd")
    import sys
    sys.path.insert(0, sys_path)
    from cd360 import job
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, num_poses, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = torch.cat([job.sample_assigned(_CpuSampler(*_job_of(p)[:3]), [_job_of(p)], steps)[0] for p in range(num_poses)])
    assert sum((m for _, m, _ in res), []) == list(range(num_poses))
    for _, _, latents in res:
        assert latents.shape == (num_poses, 4, 8, 8) and torch.equal(latents, want)
    with pytest.raises(ValueError):
        job.sample_poses(_CpuSampler, _job_of, 1, steps, world=2, rank=1)
