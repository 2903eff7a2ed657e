"""Does the feature encoder run faster in sub-batches whose intermediates fit the 256 MiB Infinity Cache?
(64 ch x 128 x 128 x 64 samples = 268 MB per tensor at the stacked batch of 64; 134 MB at 32; 67 MB at 16.)
   python tools/lab/enc_split.py [pairs]        -> feature encoder (IN) on 2 x pairs images as 1 / 2 / 4 / 8 sub-batches,
                                                  context encoder (BN) on pairs images as 1 / 2 / 4"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    model, _ = bench.build_model(8, 'cuda')
    x = torch.rand(2 * pairs, 3, 256, 256, device='cuda')
    enc, ctx = model.render_encoder, model.context
    for rep in range(2):
        for parts in (1, 2, 4, 8):
            chunks = x.chunk(parts)
            ms = timed(lambda: [enc(c) for c in chunks])
            print(f'rep {rep} feature encoder (IN), {2 * pairs} images as {parts} x {2 * pairs // parts}: {ms:7.3f} ms', flush=True)
        xc = x[:pairs]
        for parts in (1, 2, 4):
            chunks = xc.chunk(parts)
            ms = timed(lambda: [ctx(c) for c in chunks])
            print(f'rep {rep} context encoder (BN), {pairs} images as {parts} x {pairs // parts}: {ms:7.3f} ms', flush=True)


if __name__ == '__main__':
    main()
