"""Batch-N latency (hipGraph replay) with the library's automatic K slicing of small-grid convolutions on / off, alternating.
    python tools/lab/b1_autoslice_ab.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench
from scflow_amd import ops
from scflow_amd.graph import GraphedRefiner


def main():
    for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
        model, _ = bench.build_model(8, 'cuda:0')
        b = bench.make_batch(n, seed=5, device='cuda:0')
        graphs = {}
        ops.register_conv_workspace(True)
        ops.side_stream_handle()
        for mode in (1, 0):
            ops.tune('conv_autoslice', mode)
            for _ in range(3):
                bench.run_step(model, b)
            torch.cuda.synchronize()
            graphs[mode] = GraphedRefiner(model, b)
        ops.tune('conv_autoslice', 1)
        for rep in range(3):
            for mode in (1, 0):
                g = graphs[mode]
                for _ in range(3):
                    g(b)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(30):
                    g(b)
                torch.cuda.synchronize()
                print(f'batch {n} rep {rep} autoslice={mode}: hipGraph {(time.perf_counter() - t) / 30 * 1e3:.3f} ms per step', flush=True)


if __name__ == '__main__':
    main()
