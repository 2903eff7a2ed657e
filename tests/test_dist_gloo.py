"""CPU, world_size 2, gloo: the batch-split plumbing used for N > 1 GPUs
(scflow_amd/dist.py).  The HIP kernels cannot run here, so each rank 'refines' its shard
with a deterministic stand-in and the test checks partitioning, ordering and the gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scflow_amd.dist import gather_poses, shard_range


def test_shard_range_partitions():
    for total in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scflow_amd.dist import init_from_env
    r, w, _ = init_from_env('gloo')
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    rot = torch.randn((total, 3, 3), generator=g)
    tr = torch.randn((total, 3), generator=g)
    lo, hi = shard_range(total, rank, world)
    my_r, my_t = rot[lo:hi] * 2.0, tr[lo:hi] + 1.0          # stand-in for the refinement
    all_r, all_t = gather_poses(my_r, my_t, total)
    ok = torch.equal(all_r, rot * 2.0) and torch.equal(all_t, tr + 1.0)
    all_r2, _ = gather_poses(my_r, my_t)                      # total inferred by all_reduce
    ok = ok and torch.equal(all_r2, rot * 2.0)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize('total', [8, 7])
def test_two_rank_gather_gloo(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
