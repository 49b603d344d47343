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
