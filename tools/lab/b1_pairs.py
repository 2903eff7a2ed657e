"""r6: side-stream branches vs merged launches (scf_conv2d_pair) at batch N, hipGraph replays; checksums must agree.
    python tools/lab/b1_pairs.py [batch] [replays]"""
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
configs = [('streams c+f+m', {'context', 'flow', 'mask'}, set()),
           ('none', set(), set()),
           ('stream c, pairs f+m', {'context'}, {'flow', 'mask'}),
           ('pairs f+m', set(), {'flow', 'mask'}),
           ('streams c+f, pair m', {'context', 'flow'}, {'mask'}),
           ('pairs c+f+m', set(), {'context', 'flow', 'mask'}),
           ('pair c', set(), {'context'})]
for rep in range(2):
    for name, streams, pairs in configs:
        ops.OVERLAP_BRANCHES, ops.PAIR_BRANCHES = set(streams), set(pairs)
        g = GraphedRefiner(model, d)
        for k in g.static_in:
            g.static_in[k].copy_(d[k])
        for _ in range(5):
            out = g()
        torch.cuda.synchronize()
        cs = f'{float(out[0][-1].double().abs().sum()):.9e} {float(out[2][-1].double().sum()):.12f}'
        t0 = time.perf_counter()
        for _ in range(K):
            g.graph.replay()
        torch.cuda.synchronize()
        print(f'rep {rep} batch {n} {name:24s}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms  checksum {cs}', flush=True)
        del g
ops.OVERLAP_BRANCHES, ops.PAIR_BRANCHES = {'context', 'flow', 'mask'}, set()
