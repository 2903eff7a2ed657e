"""Does running two half-batches concurrently on two streams (their kernels de-synchronised: one's epilogues under
the other's MFMA loops) beat one batch-32 step?  Two model instances (own workspaces), two torch streams, one or
two host threads."""
import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
DEV = 'cuda:0'
torch.cuda.set_device(0)
m32, _ = bench.build_model(8, DEV)
b32 = bench.make_batch(32, 1000, DEV)


def rate(fn, pairs, secs=4.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        fn(); n += 1
    torch.cuda.synchronize()
    return n * pairs / (time.perf_counter() - t0)


print('1 stream x 32        :', round(rate(lambda: bench.run_step(m32, b32), 32), 1), 'pairs/s', flush=True)
for half in (16, 8):
    k = 32 // half
    ms = [bench.build_model(8, DEV)[0] for _ in range(k)]
    bs = [bench.make_batch(half, 1000 + i, DEV) for i in range(k)]
    ss = [torch.cuda.Stream() for _ in range(k)]
    print(f'1 stream x {half}        :', round(rate(lambda: bench.run_step(ms[0], bs[0]), half), 1), 'pairs/s', flush=True)

    def both():
        for m, b, s in zip(ms, bs, ss):
            with torch.cuda.stream(s):
                bench.run_step(m, b)
    print(f'{k} streams x {half}, 1 thread:', round(rate(both, 32), 1), 'pairs/s', flush=True)

    def threaded():
        def w(i):
            torch.cuda.set_device(0)
            with torch.cuda.stream(ss[i]):
                bench.run_step(ms[i], bs[i])
        ts = [threading.Thread(target=w, args=(i,)) for i in range(k)]
        [t.start() for t in ts]; [t.join() for t in ts]
    print(f'{k} streams x {half}, {k} threads:', round(rate(threaded, 32), 1), 'pairs/s', flush=True)
    del ms, bs
