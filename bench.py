#!/usr/bin/env python
"""Benchmark of the SCFlow refinement hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch per GPU: ``SCFlowRefiner.get_pose`` on B
synthetic 256x256 image pairs, 8 GRU iterations, fp32, seeded random weights of the reference
architecture (no dataset / checkpoint is reachable).  Inputs are resident in HBM before the
timed region.  Metric (BASELINE.json): image pairs per second, whole job (all GPUs).

Extra objects on the JSON line:
  roofline      corr-lookup kernel: algorithmic bytes per launch (2904 B/query) /
                average launch duration measured live with HIP events on the launch stream
                inside the timed steps, against the 8 TB/s HBM3E peak.
  cpu_baseline  the CPU oracle (oracle/, a port of the reference's torch path) timed on the
                host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOOKUP_BYTES_PER_QUERY = 2904  # 4*(10*10*4) read + 8 flow + 4*(9*9*4) write


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


def cpu_baseline(sd, iters: int, pairs: int = 8, reps: int = 10, budget_s: float = 20.0):
    """the oracle on the host cores: bounded sample (<= ``reps`` passes over ``pairs`` pairs
    after one warm-up pair, stopped once ``budget_s`` seconds of CPU work are spent)."""
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
        while done < reps and time.perf_counter() - t0 < budget_s:
            oracle.get_pose(*args, iters=iters)
            done += 1
        dt = time.perf_counter() - t0
    return dict(value=round(pairs * done / dt, 3), unit='pairs/s', cores=cores, kind='port',
                sample=f'{done} x oracle.get_pose on a batch of {pairs} synthetic 256x256 pairs, '
                       f'{iters} iters, torch CPU fp32, {cores} threads (cgroup quota), after 1 '
                       f'warm-up pair ({dt:.1f} s timed)')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='image pairs per GPU per step')
    ap.add_argument('--iters', type=int, default=8)
    ap.add_argument('--precision', choices=['f32', 'f16x3'], default='f32',
                    help='convolution arithmetic: exact fp32 MFMA, or split-fp16 3xMFMA '
                         '(fp32 accumulate, ~22 mantissa bits; see DESIGN.md)')
    ap.add_argument('--no-alt', action='store_true', help='skip the second timed loop in the other '
                    'convolution precision')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-batch1', action='store_true')
    args = ap.parse_args()

    from scflow_amd import ops
    from scflow_amd.dist import gather_poses, init_from_env
    ops.set_conv_precision(args.precision)
    rank, world, local = init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f'[bench] note: WORLD_SIZE={world} but --gpus {args.gpus}; using {world}',
                  file=sys.stderr)
    dev_index = local % torch.cuda.device_count()   # one process per GPU (torchrun LOCAL_RANK)
    torch.cuda.set_device(dev_index)
    device = f'cuda:{dev_index}'

    model, sd = build_model(args.iters, device)
    batch = make_batch(args.batch, seed=1000 + rank, device=device)

    def step():
        outs = run_step(model, batch)
        rot, trans = outs[2][-1], outs[3][-1]
        if world > 1:
            rot, trans = gather_poses(rot, trans, args.batch * world)
        return rot, trans

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(precision):
        """W warm-up + K timed steps in one convolution precision -> (seconds, lookup us list)."""
        ops.set_conv_precision(precision)
        for _ in range(args.warmup):
            step()
        fence()
        ops.lookup_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        lk = ops.lookup_timing(False)
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, lk

    dt, lookup_us = timed(args.precision)
    # Secondary measurements: never allowed to take the headline line down with them.
    conv_launches, cb_us, cb_ev = None, None, []
    try:
        # the kernels that dominate the step by TIME are the fp32 MFMA convolutions: one extra
        # (untimed) step with every conv launch bracketed by events on its stream
        ops.conv_timing(True)
        step()
        conv_launches = ops.conv_timing(False)
        # north_star: MFMA utilisation of the correlation-volume build (dense fmap1 . fmap2^T).  The
        # level-0 contraction alone (num_levels=1: no pooling cascade), same shapes as in the step.
        fa = torch.randn((args.batch, 256, 32, 32), device=device)
        fb = torch.randn((args.batch, 256, 32, 32), device=device)
        lv0 = [torch.empty((args.batch * 1024, 1, 32, 32), device=device)]
        for _ in range(3):
            ops.corr_build(fa, fb, 1, out=lv0, level0_tiled=True)
        cb_ev = [ops.time_first_kernel(lambda: ops.corr_build(fa, fb, 1, out=lv0, level0_tiled=True))
                 for _ in range(10)]
        cb_us = sum(cb_ev) / len(cb_ev)
        del fa, fb, lv0
    except Exception as exc:          # pragma: no cover - reported, not fatal
        print(f'[bench] secondary measurement failed: {exc!r}', file=sys.stderr)
    alt = None
    if not args.no_alt:
        other = 'f16x3' if args.precision == 'f32' else 'f32'
        dt_alt, lk_alt = timed(other)
        alt = {'precision': other,
               'value': round(args.batch * world * args.steps / dt_alt, 2), 'unit': 'pairs/s',
               'ms_per_step': round(dt_alt / args.steps * 1e3, 3),
               'lookup_avg_launch_us': round(sum(lk_alt) / max(len(lk_alt), 1), 2),
               'note': 'f16x3 = spatial convs with >=16 input channels as 3 fp16 MFMAs over an exact '
                       'hi/lo split of both operands, fp32 accumulate (~22 mantissa bits); flow EPE vs '
                       'the fp32 CPU oracle 7.6e-5 px over 8 iterations (tests/test_gpu_refiner.py), '
                       'north-star tolerance 1e-3 px.  f32 = v_mfma_f32_32x32x2_f32 everywhere.'}
        ops.set_conv_precision(args.precision)

    result = None
    if rank == 0:
        pmc = {}
        pmc_path = os.path.join(ROOT, 'profiles', 'lookup_pmc.json')
        if os.path.exists(pmc_path) and args.batch == 32:      # PMC passes are separate rocprofv3 runs
            pmc = json.load(open(pmc_path))
        pairs = args.batch * world * args.steps
        q = args.batch * 32 * 32
        avg_us = sum(lookup_us) / max(len(lookup_us), 1)
        achieved = LOOKUP_BYTES_PER_QUERY * q / (avg_us * 1e-6) / 1e9 if lookup_us else None
        result = {
            'metric': 'image-pairs/sec at 256x256, 8 GRU iters',
            'value': round(pairs / dt, 2), 'unit': 'pairs/s', 'n_gpus': world,
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
            'roofline': {'kernel': 'corr_lookup_kernel<4, true, 32>', 'bound': 'hbm',
                         'achieved': None if achieved is None else round(achieved, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': pmc.get('traffic_bytes_per_launch'),
                         'traffic_source': pmc.get('source'),
                         'avg_launch_us': round(avg_us, 2), 'launches_timed': len(lookup_us),
                         'algorithmic_bytes_per_launch': LOOKUP_BYTES_PER_QUERY * q},
        }
        rk = pmc.get('rocprof_kernel_trace')
        if rk:      # committed `rocprofv3 --kernel-trace --stats` pass of this command (profiles/)
            result['roofline']['rocprof_avg_launch_us'] = rk['avg_us']
            result['roofline']['note'] = (
                'avg_launch_us: HIP start/stop events bound to each lookup launch of the timed steps '
                '(hipExtLaunchKernel on the launch stream = the dispatch\'s own begin/end timestamps); '
                f"the committed kernel trace of this command averages {rk['avg_us']} us = "
                f"{LOOKUP_BYTES_PER_QUERY * q / rk['avg_us'] / 1e3 / HBM_PEAK_GBS:.3f} of peak")
        cb_fl = 2.0 * 256 * 1024 * 1024 * args.batch
        if cb_us:
          result['roofline_corr_build'] = {
            'kernel': 'conv_mfma_kernel<4, 1, 32> with per-sample weights (scf_corr_build level 0: '
                      'fmap1 . fmap2^T / sqrt(C), 8x4-tiled store)', 'bound': 'mfma',
            'achieved': round(cb_fl / (cb_us * 1e-6) / 1e12, 1), 'peak': MFMA_F32_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(cb_fl / (cb_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            'avg_launch_us': round(cb_us, 1), 'launches_timed': len(cb_ev),
            'algorithmic_flops_per_launch': cb_fl,
            'note': f'2*C*(h*w)^2 flops per pair, C=256, h=w=32, {args.batch} pairs; the launch also '
                    'writes the 4*(h*w)^2 B volume per pair'}
        if conv_launches:
            c_us = sum(u for u, _ in conv_launches)
            c_fl = sum(f for _, f in conv_launches)
            result['roofline_conv'] = {
                'kernel': 'conv_dma_kernel / conv_mfma_kernel (all convolution launches of one step)',
                'bound': 'mfma', 'achieved': round(c_fl / (c_us * 1e-6) / 1e12, 1),
                'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(c_fl / (c_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                'launches_timed': len(conv_launches), 'conv_us_per_step': round(c_us, 1),
                'share_of_step': round(c_us * 1e-6 / (dt / args.steps), 3),
                'algorithmic_flops_per_step': c_fl,
                'note': 'v_mfma_f32_32x32x2_f32 (exact fp32), dense peak 256 CU x 256 flop/clk x 2.4 GHz; '
                        'flops = 2*Cin*KH*KW*Cout*Ho*Wo*N per launch; HIP start/stop events bound to each launch'}

    # ---- config[1]: single pair latency (rank 0, informational) ----
    if rank == 0 and world == 1 and not args.no_batch1:
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
            result['batch1'] = {'workload': 'BASELINE configs[1]: batch=1, 256x256, 8 iters',
                                'ms_per_pair_hipgraph': round(ms, 3),
                                'pairs_per_s_hipgraph': round(1e3 / ms, 2),
                                'ms_per_pair_eager': round(ms_eager, 3)}
        except Exception as exc:      # pragma: no cover - informational block, never fatal
            print(f'[bench] batch-1 block failed: {exc!r}', file=sys.stderr)

    if rank == 0 and alt is not None:
        result['alt_precision'] = alt
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result['cpu_baseline'] = cpu_baseline(sd, args.iters)
        except Exception as exc:      # pragma: no cover
            print(f'[bench] cpu baseline failed: {exc!r}', file=sys.stderr)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
