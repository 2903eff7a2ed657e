"""In-pipeline A/B of the lookup kernel variants (scf_tune 'lookup_pipe'): the batch-32 step with every lookup
launch carrying its own timer, the variants alternating block by block on the same model and inputs.
    python tools/lab/lookup_inpipe.py [batch] [steps per block] [blocks] [modes, e.g. 1,2]
Prints per mode: lookup mean / median us, fraction of 8 TB/s, step ms."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench
from scflow_amd import ops


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    # modes: comma-separated 'pipe' or 'pipe:store' settings of the two lookup knobs
    modes = [tuple(int(v) for v in m.split(':')) for m in (sys.argv[4].split(',') if len(sys.argv) > 4 else ['1', '2'])]
    modes = [m if len(m) == 2 else (m[0], 0) for m in modes]
    model, _ = bench.build_model(8, 'cuda')
    d = bench.make_batch(batch, 1000, 'cuda')
    for _ in range(5):
        bench.run_step(model, d)
    torch.cuda.synchronize()
    ops.lookup_timing(True, reserve=steps * 8)
    algo = 2904.0 * batch * 1024
    for b in range(blocks):
        for m in modes:
            ops.tune('lookup_pipe', m[0])
            ops.tune('lookup_store', m[1])
            bench.run_step(model, d)
            torch.cuda.synchronize()
            ops.lookup_timing_reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                bench.run_step(model, d)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            us = ops.lookup_timing_read()
            mean, med = statistics.fmean(us), statistics.median(us)
            per = [statistics.fmean(us[i::8]) for i in range(8)] if len(us) % 8 == 0 else []
            print(f'block {b} lookup_pipe={m}: lookup mean {mean:6.2f} us median {med:6.2f} us '
                  f'frac {algo / mean / 1e6 / 8e6 * 1e6:.3f} min {min(us):.2f} max {max(us):.2f}  step {dt * 1e3:7.3f} ms '
                  f'{batch / dt:7.1f} pairs/s  per iteration ' + ' '.join(f'{v:.1f}' for v in per), flush=True)
    ops.tune('lookup_pipe', 0)
    ops.tune('lookup_store', 0)
    ops.lookup_timing(False)


if __name__ == '__main__':
    main()
