"""Micro-benchmark of the two correlation kernels (standalone, HIP-event timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scflow_amd import ops
dev = 'cuda:0'


def timeit(fn, n=50, flush=None):
    for _ in range(5):
        fn()
    tot = 0.0
    evs = []
    for _ in range(n):
        if flush is not None:
            flush.add_(1.0)          # 1 GiB read+write: evicts L2 and the 256 MiB Infinity Cache
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


for N in (1, 8, 32, 64):
    h = w = 32
    f1 = torch.randn(N, 256, h, w, device=dev); f2 = torch.randn(N, 256, h, w, device=dev)
    pyr = ops.corr_build(f1, f2, 4)
    flow = torch.randn(N, 2, h, w, device=dev) * 3
    out = ops.corr_lookup(pyr, flow, 4)
    flush = torch.zeros(256 * 1024 * 1024, device=dev)
    med, mn = timeit(lambda: ops.corr_lookup(pyr, flow, 4, out=out))
    medc, mnc = timeit(lambda: ops.corr_lookup(pyr, flow, 4, out=out), n=20, flush=flush)
    gb = 2904 * N * h * w / 1e3
    print(f'N={N:3d} lookup warm {med:7.1f} us ({gb / med:7.1f} GB/s)  cold {medc:7.1f} us ({gb / medc:7.1f} GB/s)')
    pyt = ops.corr_build(f1, f2, 4, level0_tiled=True)
    medt, _ = timeit(lambda: ops.corr_lookup(pyt, flow, 4, out=out, level0_tiled=True))
    medtc, _ = timeit(lambda: ops.corr_lookup(pyt, flow, 4, out=out, level0_tiled=True), n=20, flush=flush)
    print(f'N={N:3d} tiled  warm {medt:7.1f} us ({gb / medt:7.1f} GB/s)  cold {medtc:7.1f} us ({gb / medtc:7.1f} GB/s)')
    med, mn = timeit(lambda: ops.corr_build(f1, f2, 4, out=pyt, level0_tiled=True), n=20)
    print(f'N={N:3d} build tiled {med:7.1f} us')
    med, mn = timeit(lambda: ops.corr_build(f1, f2, 4, out=pyr), n=20)
    print(f'N={N:3d} build  {med:7.1f} us  {2 * 256 * (h * w) ** 2 * N / med / 1e6:6.2f} TFLOP/s')
    del flush
