"""r6: what does ONE cross-stream fork / join cost a hipGraph replay on this ROCm?  The batch-1 pass with every real branch off
(single-stream graph), the same with a DUMMY side branch (one tiny kernel on the side stream, forked and joined once at the start of
the pass), for 2 / 4 / 8 refinement iterations (= 3 graph sizes: does the penalty scale with the node count?).
    python tools/lab/graph_fork_penalty.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from scflow_amd import ops
from scflow_amd.graph import GraphedRefiner

ops.PAIR_BRANCHES = set()      # streams only (r6 default: merged launches)
model, _ = bench.build_model(8, 'cuda:0')
d = bench.make_batch(1, 5, 'cuda:0')
ops.OVERLAP_BRANCHES = set()
scratch = torch.zeros(64, device='cuda:0')
orig = model.get_pose


def with_dummy(*a, **k):
    ev = ops.fork_point()
    br = ops.side_stream(True, after=ev)
    with br:
        scratch.add_(1.0)
    br.join()
    return orig(*a, **k)


def with_dummy_end(*a, **k):
    ev = ops.fork_point()
    out = orig(*a, **k)
    br = ops.side_stream(True, after=ev)        # side work that depends only on the start, joined at the very end
    with br:
        scratch.add_(1.0)
    br.join()
    return out


for iters in (2, 4, 8):
    model.decoder.iters = iters
    for rep in range(2):
        for name, fn in (('single stream', orig), ('+ dummy fork/join at the start', with_dummy), ('+ dummy branch spanning the pass', with_dummy_end)):
            model.get_pose = fn
            g = GraphedRefiner(model, d)
            for k_ in g.static_in:
                g.static_in[k_].copy_(d[k_])
            for _ in range(5):
                g()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                g.graph.replay()
            torch.cuda.synchronize()
            print(f'iters {iters} rep {rep} {name:34s}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per replay', flush=True)
            del g
model.get_pose = orig
