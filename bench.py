#!/usr/bin/env python
"""Benchmark of the SCFlow refinement hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]

``--gpus N`` with N > 1 and no torchrun environment RE-LAUNCHES this script under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (one
rank per GPU, RCCL); under an existing torchrun environment (RANK / WORLD_SIZE set, the driver's
way of starting it) it just joins.  It refuses (exit code 2) when fewer than N GPUs are visible or
when WORLD_SIZE disagrees with --gpus: it never silently runs a different number of ranks.

One *step* = one pass of the hot path over one batch per GPU: ``SCFlowRefiner.get_pose`` on B
synthetic 256x256 image pairs, 8 GRU iterations, fp32, seeded random weights of the reference
architecture (no dataset / checkpoint is reachable).  Inputs are resident in HBM before the
timed region.  Metric (BASELINE.json): image pairs per second, whole job (all GPUs).

Timing: W warm-up steps (and at least ``--min-warmup-seconds``), then BLOCKS of exactly K steps,
each bracketed by barrier + synchronize on both sides and reduced with MAX over ranks, repeated
until ``--min-seconds`` (default 3 s) of timed work have run; ``value`` / ``ms_per_step`` come
from the MEDIAN block, ``spread`` carries min / max.

Extra objects on the JSON line:
  roofline      corr-lookup kernel: algorithmic bytes per launch (2904 B/query) /
                average launch duration measured live with HIP events bound to each launch on the
                launch stream inside the timed steps, against the 8 TB/s HBM3E peak.
  config4       BASELINE configs[4]: RAFTRefinerFlowMask, 480x640, 12 iterations, batch 8 -- the
                lookup on a 933 MiB pyramid (larger than the 256 MiB Infinity Cache).
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's torch path) timed on the
                host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOOKUP_BYTES_PER_QUERY = 2904  # 4*(10*10*4) read + 8 flow + 4*(9*9*4) write


# Counter summaries under profiles/ (rocprofv3 --pmc passes are separate runs) are attached to the line ONLY when they
# were measured on the kernel sources of this tree: tools/summarize_pmc.py / summarize_mfma.py record sha256 of the
# files below at collection time, kernel_source_hashes() is compared at bench time.  A content
# hash, not `git rev-parse HEAD:scflow_amd/csrc`: the GPU box holds a snapshot without .git.
LOOKUP_SOURCES = ('corr_lookup.hip', 'scf_common.h', 'scf_dma.h')
CONV_SOURCES = ('conv_wino.hip', 'conv_wino1d.hip', 'conv_wino1d4.hip', 'conv_dma.hip', 'conv_mfma.hip', 'conv_taps.hip',
                'conv_thin.hip', 'conv_kernels.h', 'conv_taps_body.h', 'fc.hip', 'corr_gemm.hip', 'scf_common.h', 'scf_dma.h')


def kernel_source_hashes(files):
    import hashlib
    out = {}
    for f in files:
        try:
            out[f] = hashlib.sha256(open(os.path.join(ROOT, 'scflow_amd', 'csrc', f), 'rb').read()).hexdigest()[:16]
        except OSError:
            out[f] = None
    return out


def _fresh(summary, files):
    """(True, None) when ``summary`` (a profiles/*.json dict) was measured on today's sources, else (False, why)."""
    have = summary.get('kernel_source_hashes') if isinstance(summary, dict) else None
    if not have:
        return False, 'the summary records no kernel_source_hashes (collected before r6)'
    now = kernel_source_hashes(files)
    changed = sorted(f for f in files if have.get(f) != now[f])
    return (not changed), (f'sources changed since the counter pass: {changed}' if changed else None)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='image pairs per GPU per step')
    ap.add_argument('--iters', type=int, default=8)
    ap.add_argument('--min-seconds', type=float, default=3.0,
                    help='repeat blocks of --steps steps until this much timed work has run')
    ap.add_argument('--min-warmup-seconds', type=float, default=1.0)
    ap.add_argument('--conv-algo', choices=['winograd', 'direct'], default='winograd',
                    help="fp32 3x3 stride-1 layers: 'winograd' = F(2x2,3x3) kernel on large grids (default), "
                         "'direct' = direct kernels everywhere (fp32 fma chains in the reference's summation order)")
    ap.add_argument('--precision', choices=['f32', 'f16x3'], default='f32',
                    help='convolution arithmetic: exact fp32 MFMA, or split-fp16 3xMFMA '
                         '(fp32 accumulate, ~22 mantissa bits; see DESIGN.md)')
    ap.add_argument('--alt', action='store_true',
                    help='also time the step with the direct kernels on every layer (alt_direct) and in the other '
                         'convolution precision (alt_precision: split-fp16, narrower than the reference\'s fp32 -- never a '
                         'headline).  Off by default since r6: the two loops cost ~8 s of the default run')
    ap.add_argument('--no-alt', action='store_true', help='(default since r6; kept so that old command lines still parse)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-batch1', action='store_true')
    ap.add_argument('--no-config4', action='store_true')
    ap.add_argument('--top-layers', type=int, default=14,
                    help='convolution layers listed in roofline_conv.top_layers')
    ap.add_argument('--share-device', action='store_true',
                    help='N ranks on ONE GPU (every rank uses cuda:0, collectives over gloo): exercises the '
                         'multi-process path -- self-launch, concurrent library load, barriers, max-over-ranks '
                         'timing, pose gather with device tensors -- where only one GPU is reachable.  The '
                         'line is marked "share_device": true and is NOT a scaling measurement')
    ap.add_argument('--dump-poses', default=None,
                    help='rank 0 writes the gathered final poses of one extra step (torch.save) here')
    ap.add_argument('--standin', action='store_true',
                    help='launcher self-test (tests/test_bench_launcher.py): the step is a CPU '
                         'stand-in, backend gloo; the line is marked "standin": true and is not a '
                         'measurement')
    return ap.parse_args(argv)


# ------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def maybe_self_launch(args) -> None:
    """--gpus N > 1 outside a torchrun environment: start N ranks of this script (one per GPU)
    and exit with the launcher's return code.  Mirrors what the reference does with
    ``tools/dist_test.sh`` -> ``torch.distributed.launch`` -> test.py:100-127."""
    if args.gpus <= 1 or 'WORLD_SIZE' in os.environ:
        return
    if not args.standin and not args.share_device:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f'[bench] --gpus {args.gpus} but only {have} GPU(s) are visible: refusing to run '
                  'a different number of ranks', file=sys.stderr)
            sys.exit(2)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this driver (RCCL)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    print(f'[bench] launching {args.gpus} ranks: {" ".join(cmd)}', file=sys.stderr)
    sys.exit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------ workload
def build_model(iters: int, device: str):
    import scflow_amd
    shapes = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'state_dict_keys.json')))['shapes']
    sd = scflow_amd.fill_state_dict(shapes, seed=0)
    model = scflow_amd.build_refiner(scflow_amd.scflow_model_cfg(iters=iters))
    model.load_state_dict(sd, strict=True)
    return model.to(device), sd


def make_batch(batch: int, seed: int, device: str):
    import scflow_amd
    inp = scflow_amd.make_inputs(batch, 256, 256, seed=seed)
    return {k: v.to(device) for k, v in inp.items()}


def run_step(model, d):
    return model.get_pose(d['render_images'], d['real_images'], d['ref_rotation'],
                          d['ref_translation'], d['depth'], d['internel_k'], d['label'])


def pin_rank_to_cores(slot: int, world: int):
    """N > 1: bind this rank's host threads to its own slice of the CPUs the process may use, so that N
    Python launch loops do not migrate over one another inside a small cgroup quota (the GPU box: 16 CPUs
    for 8 ranks).  Returns the CPU list, or None where affinity cannot be set."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        per = max(1, min(len(cpus), host_cores()) // max(world, 1))
        mine = cpus[(slot * per) % len(cpus):(slot * per) % len(cpus) + per] or cpus[:1]
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return None


def host_cores() -> int:
    """CPUs this process may actually use: min(cpu_count, affinity mask, cgroup cpu.max quota).
    (The GPU box exposes 256 logical CPUs but a 16-CPU cgroup quota; 256 torch threads on a
    16-CPU quota run ~600x slower than 16.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd, iters: int, pairs: int = 32, reps: int = 10, budget_s: float = 20.0):
    """the oracle on the host cores: bounded sample (<= ``reps`` passes over a batch of ``pairs`` pairs --
    the batch size of the GPU line -- after one warm-up pair, stopped once ``budget_s`` seconds of CPU
    work are spent; at least one pass)."""
    import torch
    import oracle
    import scflow_amd
    cores = host_cores()
    torch.set_num_threads(cores)
    inp = scflow_amd.make_inputs(pairs, 256, 256, seed=99)
    args = (inp['render_images'], inp['real_images'], inp['ref_rotation'], inp['ref_translation'],
            inp['depth'], inp['internel_k'], inp['label'], sd)
    with torch.no_grad():
        oracle.get_pose(*[a[:1] if torch.is_tensor(a) else a for a in args], iters=iters)  # warm-up
        t0 = time.perf_counter()
        done = 0
        while done < reps and (done == 0 or time.perf_counter() - t0 < budget_s):
            oracle.get_pose(*args, iters=iters)
            done += 1
        dt = time.perf_counter() - t0
    return dict(value=round(pairs * done / dt, 3), unit='pairs/s', cores=cores, kind='port',
                batch=pairs,
                sample=f'{done} x oracle.get_pose on a batch of {pairs} synthetic 256x256 pairs '
                       f'(the same call as one GPU step: same batch size, same inputs generator), '
                       f'{iters} iters, torch CPU fp32, {cores} threads (cgroup quota), after 1 '
                       f'warm-up pair ({dt:.1f} s timed)')


def _latest_mfma_pmc():
    """per-kernel SQ_VALU_MFMA_BUSY_CYCLES fractions from the newest committed counter pass
    (profiles/r*_mfma_pmc.json, written by tools/summarize_mfma.py from a rocprofv3 --pmc run of this
    command): copied into the line like roofline.traffic is, never measured live."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, 'profiles', 'r*_mfma_pmc.json'))
    if not files:
        return None
    key = lambda f: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.basename(f))]
    path = sorted(files, key=key)[-1]
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    ok, why = _fresh(d, CONV_SOURCES)
    if not ok:
        return {'stale': f'profiles/{os.path.basename(path)} NOT attached: {why}'}
    return {'source': f'profiles/{os.path.basename(path)}: ' + str(d.get('source', '')),
            'all_matrix_kernels': d.get('all_matrix_kernels', {}).get('mfma_busy_vs_chip_peak'),
            'kernels': [{'kernel': k['kernel'], 'launches': k['launches'], 'mean_duration_us': k['mean_duration_us'],
                         'mfma_busy_vs_chip_peak': k['mfma_busy_vs_chip_peak'],
                         'wait_inst_any_frac': (round(k['counters_per_launch']['SQ_WAIT_INST_ANY']
                                                      / k['counters_per_launch']['SQ_WAVE_CYCLES'], 3)
                                                if k.get('counters_per_launch', {}).get('SQ_WAVE_CYCLES') else None)}
                        for k in d.get('kernels', [])[:10]]}


def _median(xs):
    s = sorted(xs)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


class BlockTimer:
    """W warm-up steps, then blocks of exactly K steps (barrier + synchronize on both sides of
    every block, MAX over ranks per block) until ``min_seconds`` of timed work."""

    def __init__(self, step, fence, allmax, steps, warmup, min_seconds, min_warmup_seconds):
        self.step, self.fence, self.allmax = step, fence, allmax
        self.steps, self.warmup = steps, warmup
        self.min_seconds, self.min_warmup_seconds = min_seconds, min_warmup_seconds

    def run(self, before_block=None, after_block=None):
        t0 = time.perf_counter()
        done = 0
        for _ in range(self.warmup):
            self.step()
            done += 1
        # time-based part: every rank must run the SAME number of steps (a step may contain a
        # collective), so the decision to continue is itself all-reduced, once per 4 steps
        self.fence()
        while self.allmax(time.perf_counter() - t0) < self.min_warmup_seconds:
            for _ in range(4):
                self.step()
                done += 1
            self.fence()                     # also keeps the host from running far ahead
        self.fence()
        blocks, total = [], 0.0
        while True:
            if before_block:
                before_block()
            self.fence()
            t1 = time.perf_counter()
            for _ in range(self.steps):
                self.step()
            self.fence()
            el = self.allmax(time.perf_counter() - t1)
            if after_block:
                after_block()
            blocks.append(el)
            total += el
            # every rank sees the same (all-reduced) totals, so they leave the loop together
            if total >= self.min_seconds or len(blocks) >= 100000:
                break
        return blocks, done


def config4_block(device: str, pmc4=None, reps_min_s: float = 2.0):
    """BASELINE configs[4]: 480x640 crops, 12 iterations, batch 8 on the pose-free
    RAFTRefinerFlowMask route (the SCFlow pose head is hard-wired to 256x256)."""
    import torch
    import scflow_amd
    from scflow_amd import ops
    n, H, W, iters = 8, 480, 640, 12
    h, w = H // 8, W // 8
    m = scflow_amd.build_refiner(scflow_amd.raft_model_cfg(iters=iters))
    sd = scflow_amd.fill_state_dict({k: v.shape for k, v in m.state_dict().items()}, seed=9)
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    g = torch.Generator().manual_seed(3)
    rend = torch.rand((n, 3, H, W), generator=g).to(device)
    real = torch.rand((n, 3, H, W), generator=g).to(device)
    for _ in range(2):
        m.get_flow(rend, real)
    torch.cuda.synchronize()
    ops.lookup_timing(True, reserve=iters * 64)
    ops.corr_build_timing(True, reserve=64)
    t0 = time.perf_counter()
    done = 0
    while done < 3 or time.perf_counter() - t0 < reps_min_s:
        m.get_flow(rend, real)
        done += 1
        if done % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lk = ops.lookup_timing(False)
    cbs = ops.corr_build_timing(False)
    q = n * h * w
    lk_us = sum(lk) / len(lk)
    gbs = LOOKUP_BYTES_PER_QUERY * q / lk_us / 1e3
    # the correlation GEMM of the step at hw = 4800 (fused first pool; launch-bound timers)
    cb_us = sum(cbs) / len(cbs)
    cb_fl = 2.0 * 256 * (h * w) ** 2 * n
    pyr_mib = sum(n * h * w * (h >> l) * (w >> l) * 4 for l in range(4)) / 2 ** 20
    # what the kernel MOVES (PMC traffic of the committed counter pass, attached only when measured on these sources)
    # against the same time: the algorithmic 2904 B / query prices a 10 x 10 window at 400 B, the memory system
    # delivers 128-byte lines
    moved = {'frac_of_moved_bytes': None, 'moved_bytes_per_launch': None,
             'traffic_stale': True if pmc4 is None else None}
    if pmc4 and pmc4.get('traffic_bytes_per_launch'):
        tb = pmc4['traffic_bytes_per_launch']
        moved = {'moved_bytes_per_launch': tb, 'traffic': tb,
                 'traffic_over_algorithmic': round(tb / (LOOKUP_BYTES_PER_QUERY * q), 3),
                 'moved_gbs': round(tb / lk_us / 1e3, 1),
                 'frac_of_moved_bytes': round(tb / lk_us / 1e3 / HBM_PEAK_GBS, 4),
                 'fetch_bytes_per_launch': pmc4.get('fetch_bytes_per_launch'),
                 'write_bytes_per_launch': pmc4.get('write_bytes_per_launch')}
    moved['line_granularity_floor'] = {
        'wanted_bytes_per_level_window': 400, 'bytes_per_level_window_at_128B_lines': 884,
        'note': 'a 10 x 10 float window on 8 x 4-float (128-byte) tiles touches (1 + 9/8)(1 + 9/4) = 6.9 lines = 884 B for the '
                '400 B the algorithmic figure counts; no 32-float tile shape does better (16 x 2: 8.6 lines; row-major rows '
                'of 80 floats: 10-20).  The 933 MiB pyramid does not fit the 256 MiB Infinity Cache, so every line comes from '
                'HBM: frac (algorithmic bytes) is bounded by wanted / moved, frac_of_moved_bytes is what the memory system '
                'delivers (the guide quotes ~6.3 TB/s = 0.79 of spec as achievable by a copy)'}
    return {
        'workload': 'BASELINE configs[4]: RAFTRefinerFlowMask.get_flow, batch=8 synthetic 480x640 '
                    'pairs, 12 GRU iters, corr radius 4, 4 levels (pose-free route: the SCFlow pose '
                    'head is hard-wired to 256x256)',
        'value': round(n * done / dt, 2), 'unit': 'pairs/s', 'ms_per_step': round(dt / done * 1e3, 2),
        'steps': done, 'pyramid_mib': round(pyr_mib, 1),
        'roofline': {'kernel': 'corr_lookup_kernel', 'bound': 'hbm', 'achieved': round(gbs, 1),
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                     'avg_launch_us': round(lk_us, 2), 'launches_timed': len(lk),
                     'algorithmic_bytes_per_launch': LOOKUP_BYTES_PER_QUERY * q, 'queries': q, **moved},
        'roofline_corr_build': {'bound': 'mfma', 'achieved': round(cb_fl / cb_us / 1e6, 1),
                                'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                'frac': round(cb_fl / cb_us / 1e6 / MFMA_F32_PEAK_TFLOPS, 4),
                                'avg_launch_us': round(cb_us, 1), 'launches_timed': len(cbs), 'hw': h * w,
                                'kernel': 'corr_gemm_kernel<true, true> inside the step'},
        'pyramid_tiles': f'{ops.pyramid_layout(h, w, 4, 4):04b}',
    }


# ------------------------------------------------------------------ main
def main():
    args = parse_args()
    maybe_self_launch(args)

    import torch
    import torch.distributed as dist
    from scflow_amd.dist import gather_poses, init_from_env

    if args.standin or args.share_device:
        os.environ.setdefault('SCF_DIST_BACKEND', 'gloo')      # RCCL cannot put two ranks on one device
    if args.share_device:
        os.environ['LOCAL_RANK'] = '0'                          # every rank drives cuda:0
    rank, world, local = init_from_env()
    pinned = pin_rank_to_cores(local if not args.share_device else rank, world) if world > 1 else None
    if world != args.gpus:
        if rank == 0:
            print(f'[bench] WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a line '
                  'for a different number of ranks', file=sys.stderr)
        sys.exit(2)

    standin = args.standin
    if standin:
        device = 'cpu'
        dev_name = 'cpu (stand-in)'
    else:
        from scflow_amd import ops
        ops.set_conv_precision(args.precision)
        ops.set_conv_winograd(args.conv_algo == 'winograd')
        ndev = torch.cuda.device_count()
        if ndev < 1 or (world > 1 and ndev < world and not args.share_device):
            print(f'[bench] rank {rank}: {ndev} GPU(s) visible for {world} ranks', file=sys.stderr)
            sys.exit(2)
        dev_index = local            # one process per GPU (torchrun LOCAL_RANK)
        torch.cuda.set_device(dev_index)
        device = f'cuda:{dev_index}'
        dev_name = torch.cuda.get_device_name(dev_index)

    # rank -> device map, and the world size the collective backend itself reports
    rank_map = [None] * world
    coll_world = 1
    if world > 1:
        dist.all_gather_object(rank_map, {'rank': rank, 'local_rank': local, 'device': device,
                                          'name': dev_name, 'pid': os.getpid()})
        cdev = 'cpu' if dist.get_backend() == 'gloo' else device
        one = torch.ones(1, device=cdev)
        parts = [torch.zeros(1, device=cdev) for _ in range(world)]
        dist.all_gather(parts, one)
        coll_world = int(sum(float(p.item()) for p in parts))
    else:
        rank_map = [{'rank': 0, 'local_rank': local, 'device': device, 'name': dev_name,
                     'pid': os.getpid()}]

    if standin:
        model = sd = None
        g = torch.Generator().manual_seed(rank)
        a = torch.randn((args.batch, 64, 64), generator=g)

        def step():
            y = torch.bmm(a, a)
            rot = y[:, :3, :3].contiguous()
            trans = y[:, 0, :3].contiguous()
            if world > 1:
                rot, trans = gather_poses(rot, trans, args.batch * world)
            return rot, trans
    else:
        model, sd = build_model(args.iters, device)
        batch = make_batch(args.batch, seed=1000 + rank, device=device)

        def step():
            outs = run_step(model, batch)
            rot, trans = outs[2][-1], outs[3][-1]
            if world > 1:
                rot, trans = gather_poses(rot, trans, args.batch * world)
            return rot, trans

    def sync():
        if not standin:
            torch.cuda.synchronize()

    def fence():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    def allmax(x: float) -> float:
        if world > 1:
            t = torch.tensor([x], device='cpu' if dist.get_backend() == 'gloo' else device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    corr_us = []      # correlation GEMM launches of the timed steps (the kernel the step runs)

    def timed(precision, corr_out=None):
        """-> (block seconds list, lookup launch durations in us, warm-up steps done)"""
        lookups = []
        if standin:
            bt = BlockTimer(step, fence, allmax, args.steps, args.warmup, args.min_seconds,
                            args.min_warmup_seconds)
            blocks, wdone = bt.run()
            return blocks, lookups, wdone
        ops.set_conv_precision(precision)
        ops.lookup_timing(True, reserve=args.steps * args.iters)   # timers created BEFORE the loop
        ops.corr_build_timing(True, reserve=args.steps)
        bt = BlockTimer(step, fence, allmax, args.steps, args.warmup, args.min_seconds,
                        args.min_warmup_seconds)

        def before():
            ops.lookup_timing_reset()
            ops.corr_build_timing_reset()

        def after():
            lookups.extend(ops.lookup_timing_read())
            if corr_out is not None:
                corr_out.extend(ops.corr_build_timing_read())

        blocks, wdone = bt.run(before_block=before, after_block=after)
        ops.lookup_timing(False)
        ops.corr_build_timing(False)
        return blocks, lookups, wdone

    blocks, lookup_us, warm_done = timed(args.precision, corr_us)
    dt = _median(blocks)

    result = None
    conv_launches, cb_us, cb_ev = None, None, []
    alt = None
    alt_direct = None
    if not standin:
        # Secondary measurements: never allowed to take the headline line down with them.
        try:
            # the kernels that dominate the step by TIME are the fp32 MFMA convolutions: one extra
            # (untimed) step with every conv launch bracketed by events on its stream
            # (three such steps, per-launch MEDIAN: a single hiccup -- one launch of r5s took 996 us instead of 165 --
            # would otherwise sit in the per-layer table and in by_kernel)
            runs = []
            for _ in range(3):
                ops.conv_timing(True)
                step()
                runs.append(ops.conv_timing(False))
            if len({len(r) for r in runs}) == 1 and all(a[2] == b[2] for a, b in zip(runs[0], runs[1])):
                conv_launches = [(sorted(r[i][0] for r in runs)[1], runs[0][i][1], runs[0][i][2]) for i in range(len(runs[0]))]
            else:
                conv_launches = runs[-1]
            # north_star: MFMA utilisation of the correlation-volume build (dense fmap1 . fmap2^T).
            # The level-0 contraction alone, same shapes as in the step.
            fa = torch.randn((args.batch, 256, 32, 32), device=device)
            fb = torch.randn((args.batch, 256, 32, 32), device=device)
            lv0 = [torch.empty((args.batch * 1024, 1, 32, 32), device=device)]
            for _ in range(3):
                ops.corr_build(fa, fb, 1, out=lv0, tiled_levels=1)
            cb_ev = [ops.time_first_kernel(lambda: ops.corr_build(fa, fb, 1, out=lv0, tiled_levels=1))
                     for _ in range(10)]
            cb_us = sum(cb_ev) / len(cb_ev)
            del fa, fb, lv0
        except Exception as exc:          # pragma: no cover - reported, not fatal
            print(f'[bench] secondary measurement failed: {exc!r}', file=sys.stderr)
        if args.alt and args.precision == 'f32' and args.conv_algo == 'winograd':
            ops.set_conv_winograd(False)
            blocks_d, _, _ = timed('f32')
            ops.set_conv_winograd(True)
            dt_d = _median(blocks_d)
            alt_direct = {'conv_algo': 'direct', 'value': round(args.batch * world * args.steps / dt_d, 2),
                          'unit': 'pairs/s', 'ms_per_step': round(dt_d / args.steps * 1e3, 3), 'blocks': len(blocks_d),
                          'note': 'the same step with the direct kernels on every layer (ops.set_conv_winograd(False)): '
                                  'fp32 fma chains in the summation order of the reference; parity of both paths: '
                                  'tests/test_gpu_refiner.py (not measured by this run)'}
        if args.alt:
            other = 'f16x3' if args.precision == 'f32' else 'f32'
            blocks_alt, lk_alt, _ = timed(other)
            dt_alt = _median(blocks_alt)
            alt = {'precision': other,
                   'value': round(args.batch * world * args.steps / dt_alt, 2), 'unit': 'pairs/s',
                   'ms_per_step': round(dt_alt / args.steps * 1e3, 3), 'blocks': len(blocks_alt),
                   'lookup_avg_launch_us': round(sum(lk_alt) / max(len(lk_alt), 1), 2),
                   'note': 'f16x3 = spatial convs with >=16 input channels as 3 fp16 MFMAs over an exact '
                           'hi/lo split of both operands, fp32 accumulate (~22 mantissa bits); parity: '
                           'tests/test_gpu_refiner.py (not measured by this run).  f32 = v_mfma_f32_32x32x2_f32 everywhere.'}
            ops.set_conv_precision(args.precision)

    if rank == 0:
        pairs_per_block = args.batch * world * args.steps
        result = {
            'metric': 'image-pairs/sec at 256x256, 8 GRU iters',
            'value': round(pairs_per_block / dt, 2), 'unit': 'pairs/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.precision == 'f32' else 'f32 (split-fp16 3xMFMA convs, fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[2] per GPU: batch={args.batch} synthetic '
                                   f'256x256 pairs, {args.iters} GRU iters, corr radius 4, 4 levels'
                                   + (f' (configs[3] shape: {args.batch * world} pairs batch-split '
                                      f'over {world} GPUs)' if world > 1 else ''),
                       'batch_per_gpu': args.batch, 'global_batch': args.batch * world,
                       'height': 256, 'width': 256, 'iters': args.iters,
                       'parallelism': f'batch-split x{world}, no data-path collective'},
            'timing': {'blocks': len(blocks), 'steps_per_block': args.steps,
                       'timed_seconds': round(sum(blocks), 3), 'warmup_steps_run': warm_done,
                       'ms_per_step_median': round(dt / args.steps * 1e3, 3),
                       'ms_per_step_min': round(min(blocks) / args.steps * 1e3, 3),
                       'ms_per_step_max': round(max(blocks) / args.steps * 1e3, 3),
                       'note': 'value = median over blocks of exactly --steps steps (barrier + '
                               'synchronize on both sides, max over ranks per block)'},
            'ranks': rank_map, 'collective_world_size': coll_world,
        }
        if standin:
            result['standin'] = True
            result['metric'] = 'LAUNCHER SELF-TEST (CPU stand-in step, not a measurement)'
        if world > 1:
            result['host_affinity'] = {'cpus_of_rank0': pinned}
        result['scaling_note'] = ('one line = one N: no scaling curve or efficiency is reported here -- the driver '
                                  'computes it from the per-N lines of its SCALE run (configs[3] = 8 GPUs x 32 pairs)')
        if args.share_device:
            result['share_device'] = True
            result['scaling'] = 'none (ranks share one GPU)'
            result['config']['parallelism'] = (f'batch-split x{world} over ONE shared GPU (gloo): a functional run '
                                               'of the multi-process path, NOT a scaling measurement')
    if args.dump_poses:
        rot, trans = step()
        sync()
        if rank == 0:
            torch.save({'rotation': rot.cpu(), 'translation': trans.cpu(), 'world': world,
                        'batch_per_rank': args.batch, 'seeds': [1000 + r for r in range(world)]}, args.dump_poses)
    if rank == 0 and not standin:
        pmc = {}
        pmc_path = os.path.join(ROOT, 'profiles', 'lookup_pmc.json')
        pmc_all, pmc_stale = {}, None
        if os.path.exists(pmc_path):      # PMC passes are separate rocprofv3 runs
            pmc_all = json.load(open(pmc_path))
            ok, why = _fresh(pmc_all, LOOKUP_SOURCES)
            if not ok:
                pmc_all, pmc_stale = {}, why
        if args.batch == 32:
            pmc = pmc_all
        q = args.batch * 32 * 32
        avg_us = sum(lookup_us) / max(len(lookup_us), 1)
        achieved = LOOKUP_BYTES_PER_QUERY * q / (avg_us * 1e-6) / 1e9 if lookup_us else None
        result['roofline'] = {
            'kernel': pmc.get('kernel', 'corr_lookup_kernel'), 'bound': 'hbm',
            'achieved': None if achieved is None else round(achieved, 1),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
            'traffic': pmc.get('traffic_bytes_per_launch'),
            'traffic_source': pmc.get('source'),
            'avg_launch_us': round(avg_us, 2), 'launches_timed': len(lookup_us),
            'median_launch_us': round(_median(lookup_us), 2) if lookup_us else None,
            'algorithmic_bytes_per_launch': LOOKUP_BYTES_PER_QUERY * q}
        if pmc_stale:       # never repeat a counter figure of other kernel sources
            result['roofline'].update(traffic=None, traffic_stale=True, traffic_source=f'profiles/lookup_pmc.json NOT attached: {pmc_stale}')
        result['_pmc_config4'] = pmc_all.get('config4')
        rk = pmc.get('rocprof_kernel_trace')
        if rk:      # committed `rocprofv3 --kernel-trace --stats` pass of this command (profiles/)
            result['roofline']['rocprof_avg_launch_us'] = rk['avg_us']
            result['roofline']['note'] = (
                'avg_launch_us: HIP start/stop events bound to each lookup launch of the timed steps '
                '(hipExtLaunchKernel on the launch stream = the dispatch\'s own begin/end timestamps; '
                'timers are created before the timed loop); '
                f"the committed kernel trace of this command averages {rk['avg_us']} us = "
                f"{LOOKUP_BYTES_PER_QUERY * q / rk['avg_us'] / 1e3 / HBM_PEAK_GBS:.3f} of peak")
        cb_fl = 2.0 * 256 * 1024 * 1024 * args.batch
        if corr_us:
            in_us = sum(corr_us) / len(corr_us)
            result['roofline_corr_build'] = {
                'kernel': 'corr_gemm_kernel<true, true>: fmap1 . fmap2^T / sqrt(C), 8x4-tiled level-0 store, '
                          'first 2x2 pool fused -- the launch the step runs',
                'bound': 'mfma',
                'achieved': round(cb_fl / (in_us * 1e-6) / 1e12, 1), 'peak': MFMA_F32_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(cb_fl / (in_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                'avg_launch_us': round(in_us, 1), 'launches_timed': len(corr_us),
                'median_launch_us': round(_median(corr_us), 1),
                'algorithmic_flops_per_launch': cb_fl,
                'note': f'2*C*(h*w)^2 flops per pair, C=256, h=w=32, {args.batch} pairs; HIP start/stop events '
                        'bound to the GEMM launch of every timed step (one per step); the launch also writes '
                        'the 4*(h*w)^2 B volume and its first pooled level per pair'}
            if cb_us:      # the same contraction alone (no pool), launched back to back on a warm cache
                result['roofline_corr_build']['standalone_level0_only'] = {
                    'kernel': 'corr_gemm_kernel<true, false>', 'avg_launch_us': round(cb_us, 1),
                    'launches_timed': len(cb_ev),
                    'achieved': round(cb_fl / (cb_us * 1e-6) / 1e12, 1),
                    'frac': round(cb_fl / (cb_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)}
        if conv_launches:
            c_us = sum(e[0] for e in conv_launches)
            c_fl = sum(e[1] for e in conv_launches)
            w_us = sum(e[0] for e in conv_launches if e[2].endswith('[winograd]'))
            w_fl = sum(e[1] for e in conv_launches if e[2].endswith('[winograd]'))
            v_us = sum(e[0] for e in conv_launches if e[2].endswith('[winograd F(2,5)]'))
            v_fl = sum(e[1] for e in conv_launches if e[2].endswith('[winograd F(2,5)]'))
            u_us = sum(e[0] for e in conv_launches if e[2].endswith('[winograd F(4,5)]'))
            u_fl = sum(e[1] for e in conv_launches if e[2].endswith('[winograd F(4,5)]'))
            x_fl = c_fl - w_fl - v_fl - u_fl + w_fl / 2.25 + v_fl * 0.6 + u_fl * 0.4     # flops the matrix cores actually execute
            by_shape = {}
            for us, fl, tag in conv_launches:
                a = by_shape.setdefault(tag, [0, 0.0, 0.0])
                a[0] += 1; a[1] += us; a[2] += fl
            top = sorted(by_shape.items(), key=lambda kv: -kv[1][1])[:args.top_layers]
            x_tf = x_fl / (c_us * 1e-6) / 1e12
            w_x, v_x, u_x = w_fl / 2.25, v_fl * 0.6, u_fl * 0.4
            d_us, d_fl = c_us - w_us - v_us - u_us, c_fl - w_fl - v_fl - u_fl
            result['roofline_conv'] = {
                'kernel': 'conv_wino_kernel / conv_wino1d4_kernel / conv_wino1d_kernel / conv_dma_kernel / conv_mfma_kernel / conv_taps_kernel (all convolution '
                          'launches of one step)',
                # the roofline fraction of THIS line: MFMA flops the kernels actually issue / time / dense fp32 peak
                'bound': 'mfma', 'achieved': round(x_tf, 1),
                'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(x_tf / MFMA_F32_PEAK_TFLOPS, 4),
                'flops_per_step': x_fl,
                'launches_timed': len(conv_launches), 'conv_us_per_step': round(c_us, 1),
                'share_of_step': round(c_us * 1e-6 / (dt / args.steps), 3),
                'by_kernel': {
                    'winograd_f2x2_3x3': {'us_per_step': round(w_us, 1),
                                          'tflops': round(w_x / max(w_us, 1e-9) / 1e6, 1),
                                          'frac': round(w_x / max(w_us, 1e-9) / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)},
                    'winograd_f2_5': {'us_per_step': round(v_us, 1),
                                      'tflops': round(v_x / max(v_us, 1e-9) / 1e6, 1),
                                      'frac': round(v_x / max(v_us, 1e-9) / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)},
                    'winograd_f4_5': {'us_per_step': round(u_us, 1),
                                      'tflops': round(u_x / max(u_us, 1e-9) / 1e6, 1),
                                      'frac': round(u_x / max(u_us, 1e-9) / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)},
                    'direct': {'us_per_step': round(d_us, 1),
                               'tflops': round(d_fl / max(d_us, 1e-9) / 1e6, 1),
                               'frac': round(d_fl / max(d_us, 1e-9) / 1e6 / MFMA_F32_PEAK_TFLOPS, 4)}},
                # side figure, NOT a roofline fraction: the same time priced at the flops a direct convolution
                # of these layers would execute (Winograd issues 1 / 2.25 resp. 0.6 resp. 0.4 of them)
                'algorithmic': {'flops_per_step': c_fl, 'tflops': round(c_fl / (c_us * 1e-6) / 1e12, 1),
                                'winograd_f2x2_3x3_tflops': round(w_fl / max(w_us, 1e-9) / 1e6, 1),
                                'winograd_f2_5_tflops': round(v_fl / max(v_us, 1e-9) / 1e6, 1),
                                'winograd_f4_5_tflops': round(u_fl / max(u_us, 1e-9) / 1e6, 1),
                                'note': 'direct-convolution flops 2*Cin*KH*KW*Cout*Ho*Wo*N of every launch / time: what '
                                        'the layers cost in the formulation of the reference; can exceed the MFMA peak '
                                        'because the Winograd kernels do not execute these flops'},
                'top_layers': [{'layer': k, 'launches': v[0], 'us': round(v[1], 1),
                                'tflops_algorithmic': round(v[2] / v[1] / 1e6, 1),
                                'tflops': round(v[2] / v[1] / 1e6 / (2.25 if k.endswith('[winograd]') else
                                                                     (1 / 0.6) if k.endswith('[winograd F(2,5)]') else
                                                                     2.5 if k.endswith('[winograd F(4,5)]') else 1.0), 1)}
                               for k, v in top],
                'note': 'v_mfma_f32_32x32x2_f32 (fp32 throughout), dense peak 256 CU x 256 flop/clk x 2.4 GHz; achieved = '
                        'EXECUTED MFMA flops of all convolution launches of one step (direct launches 2*Cin*KH*KW*Cout*Ho*Wo*N; '
                        'F(2x2,3x3) launches 1 / 2.25 of that: 16 multiplies per 2x2 outputs instead of 36; F(2,5) launches '
                        '0.6: 6 per 2 outputs instead of 10; F(4,5) launches 0.4: 8 per 4 outputs instead of 20) / sum of the launch durations (HIP start/stop events bound to '
                        'each launch)'}
            mp = _latest_mfma_pmc()
            if mp and mp.get('stale'):
                result['roofline_conv']['mfma_busy'] = None
                result['roofline_conv']['mfma_busy_stale'] = mp['stale']
            elif mp and args.batch == 32:       # SQ counter pass of this command (own rocprofv3 run), committed under profiles/
                result['roofline_conv']['mfma_busy'] = mp
            # GRU context hoisting: the context channels' part of the SepConvGRU
            # convolutions runs once per pair instead of once per iteration.  `achieved` counts the
            # flops actually executed; the same step in the reference's formulation would execute
            # `saved` more -- reported beside it, not in place of it.
            dec = getattr(model, 'decoder', None)
            if dec is not None and getattr(dec, 'hoist_context', False):
                hc_, cc_ = dec.h_channels, dec.cxt_channels
                taps = sum(c.conv.kernel_size[0] * c.conv.kernel_size[1] for c in dec.gru.conv_z)
                saved = 2.0 * cc_ * taps * 3 * hc_ * (32 * 32) * args.batch * (args.iters - 1)
                result['roofline_conv']['gru_context_hoisting'] = {
                    'flops_saved_per_step': saved,
                    'tflops_algorithmic_in_reference_formulation': round((c_fl + saved) / (c_us * 1e-6) / 1e12, 1),
                    'note': 'conv([h|c|x]) = conv([h|x]) + conv_c(c), c = context features (constant over '
                            'the iterations): conv_c(c) once per pair; tflops_in_reference_formulation = '
                            '(executed + saved flops) / conv time, for comparison with a per-iteration GRU'}

    # ---- config[1]: single pair latency (rank 0, informational) ----
    if rank == 0 and world == 1 and not standin and not args.no_batch1:
        try:
            from scflow_amd.graph import GraphedRefiner
            b1 = make_batch(1, seed=5, device=device)
            for _ in range(3):
                run_step(model, b1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n1 = 10
            for _ in range(n1):
                run_step(model, b1)
            torch.cuda.synchronize()
            ms_eager = (time.perf_counter() - t1) / n1 * 1e3
            graphed = GraphedRefiner(model, b1)          # whole pass as one hipGraph
            for _ in range(3):
                graphed(b1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n1 = 30
            for _ in range(n1):
                graphed(b1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / n1 * 1e3
            # the same replay with the inputs already IN the graph's input buffers (a caller that produces its crops on
            # the device writes them there): no input copies, one graph launch per pair
            for k in graphed.static_in:
                graphed.static_in[k].copy_(b1[k])
            for _ in range(3):
                graphed()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n1):
                graphed()
            torch.cuda.synchronize()
            ms_inplace = (time.perf_counter() - t1) / n1 * 1e3
            result['batch1'] = {'workload': 'BASELINE configs[1]: batch=1, 256x256, 8 iters',
                                'ms_per_pair_hipgraph': round(ms, 3),
                                'pairs_per_s_hipgraph': round(1e3 / ms, 2),
                                'ms_per_pair_hipgraph_inputs_in_place': round(ms_inplace, 3),
                                'ms_per_pair_eager': round(ms_eager, 3),
                                'note': 'ms_per_pair_hipgraph includes seven device-to-device copies of the inputs into the '
                                        "graph's input buffers per pair; _inputs_in_place = the replay alone"}
            del graphed
        except Exception as exc:      # pragma: no cover - informational block, never fatal
            print(f'[bench] batch-1 block failed: {exc!r}', file=sys.stderr)

    if rank == 0 and world == 1 and not standin and not args.no_config4:
        try:
            ops.set_conv_precision(args.precision)
            result['config4'] = config4_block(device, result.get('_pmc_config4'))
        except Exception as exc:      # pragma: no cover
            print(f'[bench] config4 block failed: {exc!r}', file=sys.stderr)

    if rank == 0 and alt is not None:
        result['alt_precision'] = alt
        if alt_direct:
            result['alt_direct'] = alt_direct
    if rank == 0 and world == 1 and not standin and not args.no_cpu_baseline:
        try:
            result['cpu_baseline'] = cpu_baseline(sd, args.iters, pairs=args.batch)
        except Exception as exc:      # pragma: no cover
            print(f'[bench] cpu baseline failed: {exc!r}', file=sys.stderr)
    if rank == 0:
        result.pop('_pmc_config4', None)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
