"""Step time (batch 32 by default) under each value of one scf_tune knob, alternated:
    python tools/lab/tune_ab.py <key> <v0,v1,...> [batch] [steps] [reps]
e.g. tune_ab.py wino_variant 2,4   (quarter-domain F(2x2,3x3) kernel with 3 / 4 ring slots)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench
from scflow_amd import ops


def main():
    key = sys.argv[1]
    vals = [int(v) for v in sys.argv[2].split(',')]
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    model, _ = bench.build_model(8, 'cuda')
    d = bench.make_batch(batch, 1000, 'cuda')
    prev = ops.tune(key, vals[0])
    graph = os.environ.get('GRAPH') == '1'      # batch 1-4: time hipGraph replays (one capture per knob value)
    try:
        for rep in range(reps):
            for v in vals:
                ops.tune(key, v)
                if graph:
                    from scflow_amd.graph import GraphedRefiner
                    g = GraphedRefiner(model, d)
                    step = lambda: g(d)
                else:
                    step = lambda: bench.run_step(model, d)
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                print(f'rep {rep} {key} = {v}: {dt * 1e3:7.3f} ms per step  {batch / dt:7.1f} pairs/s', flush=True)
    finally:
        ops.tune(key, prev)


if __name__ == '__main__':
    main()
