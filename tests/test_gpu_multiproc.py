"""GPU: the N > 1 path on real hardware as far as ONE GPU allows.

``bench.py --gpus N --share-device`` (N = 2 and N = 8) self-launches N ranks (torch.distributed.run, gloo), both on
cuda:0: two processes load libscflow_hip.so side by side, run the REAL get_pose step on their own 16
pairs, meet at the barriers, reduce the block time with MAX over ranks and gather the poses (device
tensors) into job order.  The gathered poses must equal a single-process run of the same two shards
bit for bit.  This is a functional test of the multi-process path, not a scaling measurement
(configs[3] needs eight GPUs: the driver's SCALE run)."""
import json
import os
import subprocess
import sys

import pytest
import torch

import bench
import scflow_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('world,batch', [(2, 16), (8, 4), (8, 32)])
def test_ranks_share_one_gpu(tmp_path, world, batch):
    """world = 2: two shards of 16 (the r3 case).  world = 8: configs[3]'s process count on the one GPU this
    box has -- eight concurrent library loads, eight sets of >64 KB LDS attribute raises, event / timer pools and
    side streams, eight host launch loops pinned to their own CPUs inside the cgroup quota, eight-way barriers,
    MAX-over-ranks timing and pose gather.  Functional only.
    world = 8, batch = 32 (r6): **configs[3] at its stated size** -- 256 pairs, contiguous shards of 32 over 8 ranks, 8
    iterations -- as far as one GPU allows: the eight processes share cuda:0.  The gathered (256, 12) poses equal the
    single-process runs of the eight shards bit for bit, and two of the shards (ranks 3 and 7; rank 0's batch is
    test_config2_full_size_vs_oracle's) are checked against the CPU oracle.  NB each shard is decoded with ITS OWN
    label[0] (pose_head.py:209-210): a batch split equals the reference under DDP, not one batch of 256."""
    dump = os.path.join(tmp_path, 'poses.pt')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--share-device', '--batch', str(batch),
           '--steps', '2', '--warmup', '1', '--min-seconds', '0.2', '--min-warmup-seconds', '0.1', '--no-alt',
           '--no-batch1', '--no-config4', '--no-cpu-baseline', '--dump-poses', dump]
    env = dict(os.environ)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None); env.pop('LOCAL_RANK', None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == world and line['collective_world_size'] == world and line['share_device'] is True
    assert line['config']['global_batch'] == world * batch and line['value'] > 0
    pids = {r['pid'] for r in line['ranks']}
    assert len(pids) == world and {r['device'] for r in line['ranks']} == {'cuda:0'}
    assert [r['rank'] for r in line['ranks']] == list(range(world))
    assert 'NOT a scaling' in line['config']['parallelism'] and 'scaling_note' in line
    got = torch.load(dump)
    assert got['rotation'].shape == (world * batch, 3, 3) and got['translation'].shape == (world * batch, 3)
    assert got['seeds'] == [1000 + r for r in range(world)]
    # the same two shards in ONE process (each shard is its own batch: the pose head decodes a batch
    # with class label[0], pose_head.py:209-210, so the shards must not be merged)
    model, _ = bench.build_model(8, 'cuda:0')
    rots, trs = [], []
    for seed in got['seeds']:
        outs = bench.run_step(model, bench.make_batch(batch, seed, 'cuda:0'))
        rots.append(outs[2][-1].cpu()); trs.append(outs[3][-1].cpu())
    assert torch.equal(got['rotation'], torch.cat(rots)) and torch.equal(got['translation'], torch.cat(trs))
    if batch == 32:
        import json as _json
        import oracle
        shapes = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_keys.json')))['shapes']
        sd = scflow_amd.fill_state_dict(shapes, seed=0)
        torch.set_num_threads(bench.host_cores())
        for r in (3, 7):
            inp = scflow_amd.make_inputs(batch, 256, 256, seed=got['seeds'][r])
            with torch.no_grad():
                want = oracle.get_pose(inp['render_images'], inp['real_images'], inp['ref_rotation'], inp['ref_translation'],
                                       inp['depth'], inp['internel_k'], inp['label'], sd, iters=8)
            from scflow_amd.dist import shard_range
            lo, hi = shard_range(world * batch, r, world)
            er = float((got['rotation'][lo:hi] - want[2][-1]).abs().max())
            et = float((got['translation'][lo:hi] - want[3][-1]).abs().max())
            print(f'[measured] configs[3] shard {r} of 8 (pairs {lo}..{hi - 1}) final pose vs oracle: max |dR| {er:.2e}, max |dt| {et:.2e} mm')
            assert er <= 2e-5 and et <= 1e-2
