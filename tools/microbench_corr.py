"""Micro-benchmark of the two correlation kernels (standalone, HIP-event timed): row-major pyramid vs
the tile masks of ``ops.pyramid_layout`` at BASELINE configs[2] (32 x 32x32 maps) and configs[4]
(8 x 60x80 maps).   python tools/microbench_corr.py [N h w]..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scflow_amd import ops
dev = 'cuda:0'


def timeit(fn, n=50, flush=None):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)          # 1 GiB read+write: evicts L2 and the 256 MiB Infinity Cache
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


args = [int(a) for a in sys.argv[1:]]
cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(32, 32, 32), (8, 60, 80)]
flush = torch.zeros(256 * 1024 * 1024, device=dev)
for N, h, w in cases:
    f1 = torch.randn(N, 256, h, w, device=dev); f2 = torch.randn(N, 256, h, w, device=dev)
    flow = torch.randn(N, 2, h, w, device=dev) * 3
    gb = 2904 * N * h * w / 1e3
    pref = ops.pyramid_layout(h, w, 4, 4)
    for mask in sorted({0, pref & 1, pref}):
        pyr = ops.corr_build(f1, f2, 4, tiled_levels=mask)
        out = ops.corr_lookup(pyr, flow, 4, tiled_levels=mask)
        med, _ = timeit(lambda: ops.corr_lookup(pyr, flow, 4, out=out, tiled_levels=mask))
        medc, _ = timeit(lambda: ops.corr_lookup(pyr, flow, 4, out=out, tiled_levels=mask), n=20, flush=flush)
        medb, _ = timeit(lambda: ops.corr_build(f1, f2, 4, out=pyr, tiled_levels=mask), n=20)
        print(f'N={N:3d} {h}x{w} tiles {mask:04b}: lookup warm {med:7.1f} us ({gb / med:7.1f} GB/s)  cold {medc:7.1f} us '
              f'({gb / medc:7.1f} GB/s)   build {medb:7.1f} us ({2 * 256 * (h * w) ** 2 * N / medb / 1e6:6.2f} TFLOP/s incl. pools)')
        del pyr, out
