"""r6: what each side-stream branch buys at batch N as hipGraph replays (and eagerly): all branches, each one off, none.
(rocprofv3's own per-launch host cost makes a TRACED replay host-bound, so the kernel trace cannot tell whether two long
branches overlap; wall time of untraced replays can.)   python tools/lab/b1_branches.py [batch] [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from scflow_amd import ops
from scflow_amd.graph import GraphedRefiner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ops.PAIR_BRANCHES = set()      # streams only (r6 default: merged launches)
model, _ = bench.build_model(8, 'cuda:0')
d = bench.make_batch(n, 5, 'cuda:0')
ALL = {'context', 'flow', 'mask', 'upsample'}
import itertools
configs = [('+'.join(sorted(c)) or 'none', set(c)) for k in range(5) for c in itertools.combinations(sorted(ALL), k)]
if os.environ.get('ONLY'):
    configs = [c for c in configs if c[0] in os.environ['ONLY'].split(',')]
for rep in range(2):
    for name, br in configs:
        ops.OVERLAP_BRANCHES = set(br)
        g = GraphedRefiner(model, d)
        for k in g.static_in:
            g.static_in[k].copy_(d[k])
        for _ in range(5):
            g()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            g.graph.replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        me = float('nan')
        print(f'rep {rep} batch {n} branches {name:28s}: hipGraph {ms:.3f} ms, eager {me:.3f} ms', flush=True)
        del g
ops.OVERLAP_BRANCHES = ALL
