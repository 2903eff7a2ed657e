"""Multi-GPU execution of the refinement path: an embarrassingly parallel batch split.

Every image pair is independent end to end (no cross-sample op at inference: BatchNorm runs
on running statistics, InstanceNorm is per sample), so the N pairs of a
job are partitioned into contiguous index ranges, one range per rank = one process per GPU.
There is NO collective inside the hot path; the only exchange is one ``all_gather`` of the
(9 + 3) pose floats per sample at the end (48 B/sample -- latency only on xGMI).
``torch.distributed`` is used as shipped: backend "nccl" (= RCCL) on GPUs, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

__all__ = ['shard_range', 'init_from_env', 'gather_poses']


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) slice of ``total`` samples owned by ``rank`` (sizes differ by <= 1)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f'bad rank/world {rank}/{world}')
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the default
    process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('SCF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_poses(rotation: torch.Tensor, translation: torch.Tensor, total: Optional[int] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """all_gather the per-rank (n_r,3,3) / (n_r,3) poses into job order.  Ranks may own
    different sample counts (``shard_range``): shards are padded to the largest one."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rotation, translation
    world = dist.get_world_size()
    n_r = rotation.shape[0]
    dev = rotation.device
    # gloo gathers host tensors only (RCCL/"nccl" takes device tensors): stage through the host there
    cdev = torch.device('cpu') if dist.get_backend() == 'gloo' else dev
    if total is None:
        cnt = torch.tensor([n_r], device=cdev, dtype=torch.int64)
        dist.all_reduce(cnt)
        total = int(cnt.item())
    per = (total + world - 1) // world
    packed = torch.zeros((per, 12), dtype=torch.float32, device=dev)
    packed[:n_r, :9] = rotation.reshape(n_r, 9)
    packed[:n_r, 9:] = translation
    packed = packed.to(cdev)
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed)
    rows = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        rows.append(parts[r][:hi - lo])
    allp = torch.cat(rows, 0).to(dev)
    return allp[:, :9].reshape(-1, 3, 3), allp[:, 9:]
