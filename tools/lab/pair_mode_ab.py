"""r6: scf_conv2d_pair rules (scf_tune conv_pair 0 / 1) at batch N, hipGraph replays, merged launches for all three branches.
    python tools/lab/pair_mode_ab.py [batch] [replays]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from scflow_amd import ops
from scflow_amd.graph import GraphedRefiner
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
model, _ = bench.build_model(8, 'cuda:0')
d = bench.make_batch(n, 5, 'cuda:0')
ops.OVERLAP_BRANCHES, ops.PAIR_BRANCHES = set(), {'context', 'flow', 'mask'}
for rep in range(2):
    for mode in (0, 1):
        ops.tune('conv_pair', mode)
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
        print(f'rep {rep} batch {n} conv_pair {mode}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms', flush=True)
        del g
ops.tune('conv_pair', 0)
