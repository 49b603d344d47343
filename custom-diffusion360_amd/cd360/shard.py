"""Multi-GPU sharding of the sampling job.

The path shards over independent units -- target poses / diffusion samples -- with replicated weights and `references`
and NO data-path collective; the single exchange is an all-gather of the final latents `[P, 4, L, L]` at the end of the job
(RCCL over xGMI on GPUs: backend "nccl"; the same code runs on gloo for the CPU tests).  The reference processes the poses
sequentially on one GPU (sample.py:331-349).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def assign_poses(num_poses: int, world: int, rank: int) -> List[int]:
    """Contiguous block partition: rank r gets poses [r*q + min(r, rem), ...) -- sizes differ by at most one."""
    q, rem = divmod(num_poses, world)
    start = rank * q + min(rank, rem)
    return list(range(start, start + q + (1 if rank < rem else 0)))


def gather_latents(local: torch.Tensor, num_poses: int) -> torch.Tensor:
    """local [p_local, 4, L, L] (this rank's poses, in `assign_poses` order) -> [num_poses, 4, L, L] on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    counts = [len(assign_poses(num_poses, world, r)) for r in range(world)]
    pmax = max(counts)
    pad = torch.zeros(pmax, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)
