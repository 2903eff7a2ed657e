"""Branch-level concurrency (side stream) at LARGE batch: step time with each independent branch of the step on the side
stream, one at a time and together.   python tools/lab/overlap_b32.py [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench
from scflow_amd import ops


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    model, _ = bench.build_model(8, 'cuda')
    d = bench.make_batch(batch, 1000, 'cuda')
    base = dict(ops.OVERLAP_MAX_PIXELS)
    big = 1 << 40
    variants = [('none', {}), ('flow', {'flow': big}), ('mask', {'mask': big}), ('upsample', {'upsample': big}),
                ('context', {'context': big}), ('flow+mask+upsample', {'flow': big, 'mask': big, 'upsample': big}),
                ('all', {k: big for k in base})]
    for rep in range(2):
        for name, over in variants:
            ops.OVERLAP_MAX_PIXELS.update(base)
            ops.OVERLAP_MAX_PIXELS.update(over)
            for _ in range(3):
                bench.run_step(model, d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                bench.run_step(model, d)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print(f'rep {rep} overlap {name:20s}: {dt * 1e3:7.3f} ms per step  {batch / dt:7.1f} pairs/s', flush=True)
    ops.OVERLAP_MAX_PIXELS.update(base)


if __name__ == '__main__':
    main()
